#!/usr/bin/env python3
"""Bank model of the LDS access patterns of conv3_wgrad_ws_kernel (csrc/conv_bf.hip; round 6): extra LDS-array cycles per brick from
bank conflicts, following MI355X_MICROARCH.md's LDS table (4-byte accesses: two groups of 32 lanes, bank = word mod 32, one extra
cycle per extra distinct dword on a bank; ds_read_b128: four groups of 16 lanes, bank = word mod 64), for
  xw  the producers' 4-byte stores of the transposed x image (one brick's new ring planes),
  dw  their stores of the dz image,
  ar  the consumers' A-fragment reads (five dwords per lane, read with 4-byte instructions),
  br  the consumers' B-fragment reads (ds_read_b128),
as a function of the plane / row pitches.  It predicted the measured drop (A reads 4480 -> 1600 at a channel-plane pitch of 2
instead of 4 mod 32 words; SQ_LDS_BANK_CONFLICT 58 -> 32 % of SQ_LDS_IDX_ACTIVE) -- profiles/r6k_wgrad_ring_pitch_ab.txt.
usage: python tools/lds_bank_model.py"""
import itertools
def conf_b32(addrs):  # list of 64 byte addresses (or None) -> extra cycles (2 groups of 32)
    extra=0
    for g in (range(0,32),range(32,64)):
        banks={}
        for l in g:
            a=addrs[l]
            if a is None: continue
            banks.setdefault((a//4)%32,set()).add(a//4)
        m=max([len(s) for s in banks.values()],default=1)
        extra+=m-1
    return extra
G128=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
G128=G128+[[x+32 for x in g] for g in G128]
def conf_b128(addrs):
    extra=0
    for g in G128:
        banks={}
        for l in g:
            a=addrs[l]
            for k in range(4): banks.setdefault((a//4+k)%64,set()).add(a//16)
        extra+=max(len(s) for s in banks.values())-1
    return extra
def sim(XP,DP,CO=32):
    WHY=6;ZS=288
    tot={}
    # x writes: 8 producer waves x (4 j) per term ; ring base 0, set 0 (lz=2,3)
    e_x=0
    for wv in range(8):
        ad=[[None]*64 for _ in range(4)]
        for lane in range(64):
            e=wv*64+lane
            if e>=432: continue
            rowh=e//36; rem=e%36; cpart=rem//9; pr=rem%9; lzh=rowh//6; ly=rowh%6; lz=2+lzh
            for j in range(4):
                ad[j][lane]=(4*cpart+j)*XP+((0+lz)&7)*ZS+(ly*24+2*pr)*2
        for j in range(4): e_x+=conf_b32(ad[j])
    tot['xw']=e_x*2  # two terms
    e_d=0
    DI=2 if CO==32 else 4
    for i in range(DI):
        for wv in range(8):
            ad=[[None]*64 for _ in range(4)]
            for lane in range(64):
                e=wv*64+lane+i*512
                q=(e&3)+4*(e>>8); pv=(e>>2)&63
                lx=(pv%8)*2; ly=(pv//8)%4; lz=pv//32
                for j in range(4):
                    ad[j][lane]=(4*q+j)*DP+((lz*4+ly)*16+lx)*2
            for j in range(4): e_d+=conf_b32(ad[j])
    tot['dw']=e_d*2
    # consumer A reads: 5 slots (waves 0-4 one read of tile0; waves 5-7 two tiles) ; dword reads k=0..4 per term ; 8 rows
    e_a=0; n_a=0
    for slot in range(5):
        for zz in range(2):
          for yy in range(4):
            for k in range(5):
                ad=[None]*64
                for lane in range(64):
                    li=lane&31; lh=lane>>5; c=li%16; t9=slot*2+li//16
                    if t9<9:
                        kz=t9//3; ky=t9%3
                        ad[lane]=c*XP+ky*48+16*lh+((0+zz+kz)&7)*ZS+yy*48+4*k
                    else:
                        ad[lane]=16*XP+16*lh+((zz)&7)*ZS+yy*48+4*k
                e_a+=conf_b32(ad); n_a+=1
    # slots read by waves 0-4 once and by waves 5-7 once more (kx=2 tiles): x2, two terms x2
    tot['ar']=e_a*2*2
    e_b=0
    for row in range(8):
        ad=[None]*64
        for lane in range(64):
            li=lane&31; lh=lane>>5
            ad[lane]=li*DP+16*lh+row*32
        e_b+=conf_b128(ad)
    tot['br']=e_b*2*(CO//32)*8  # terms, N tiles, 8 waves
    return tot
base=2304
for pad in (16,8,24,40,72,136,264):
    for DP in (272,):
        t=sim(base+pad,DP); print('XP pad',pad,'words%32',((base+pad)//4)%32,'DP',DP,t,'sum',sum(t.values()))
for DP in (264,272,280,288,304,320):
    t=sim(base+8,DP); print('DP',DP,'words%32',(DP//4)%32,t,'sum',sum(t.values()))

print("---- generalised: row pitch RW words, ring-plane pitch ZSW words, channel-plane pitch XPW words")
def sim2(RW,ZSW,XPW,DP=272):
    tot={}
    e_x=0
    for wv in range(8):
        ad=[[None]*64 for _ in range(4)]
        for lane in range(64):
            e=wv*64+lane
            if e>=432: continue
            rowh=e//36; rem=e%36; cpart=rem//9; pr=rem%9; lzh=rowh//6; ly=rowh%6; lz=2+lzh
            for j in range(4):
                ad[j][lane]=4*((4*cpart+j)*XPW+((0+lz)&7)*ZSW+ly*RW+pr)
        for j in range(4): e_x+=conf_b32(ad[j])
    tot['xw']=e_x*2
    e_a=0
    for slot in range(5):
        for zz in range(2):
          for yy in range(4):
            for k in range(5):
                ad=[None]*64
                for lane in range(64):
                    li=lane&31; lh=lane>>5; c=li%16; t9=slot*2+li//16
                    if t9<9:
                        kz=t9//3; ky=t9%3
                        ad[lane]=4*(c*XPW+ky*RW+4*lh+((0+zz+kz)&7)*ZSW+yy*RW+k)
                    else:
                        ad[lane]=4*(16*XPW+4*lh+((zz)&7)*ZSW+yy*RW+k)
                e_a+=conf_b32(ad)
    tot['ar']=e_a*4
    return tot
best=[]
for RW in (10,11,12,13):
    for zp in range(0,4):
        ZSW=6*RW+zp
        for xp in range(0,32):
            XPW=8*ZSW+xp
            if XPW%2: continue   # keep 8-byte alignment of planes
            t=sim2(RW,ZSW,XPW); best.append((sum(t.values()),RW,ZSW,XPW,t))
best.sort(key=lambda b:b[0])
for b in best[:12]: print(b)
print('current', sim2(12,72,578))
print('pitch 577', sim2(12,72,577), 'pitch 580', sim2(12,72,580), 'pitch 578', sim2(12,72,578))
