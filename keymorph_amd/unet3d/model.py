"""UNet3D / TruncatedUNet3D with the reference's constructor surface and state_dict keys
(keymorph/unet3d/model.py:14-189, 307-430; buildingblocks.py:10-208, 321-619), computed by the
HIP backbone operators on NDHWC activations.

``nn.GroupNorm`` / ``nn.Conv3d`` objects are used ONLY as parameter holders (identical names,
shapes and default initialisation => reference / BrainMorph checkpoints load with
``strict=True``, also under ``nn.DataParallel``'s ``module.`` prefix); their own forward is
never called.  Only the "gcr" DoubleConv configuration the reference's scripts build
(scripts/run.py:350-387) is implemented.
"""
from __future__ import annotations

import torch
import torch.nn as nn

from .. import backbone_ops as B


def number_of_features_per_level(init_channel_number, num_levels):
    return [init_channel_number * 2 ** k for k in range(num_levels)]


class SingleConv(nn.Module):
    """GroupNorm -> Conv3d(3, pad 1, no bias) -> ReLU ('gcr')."""

    def __init__(self, in_channels, out_channels, num_groups=8, first_layer=False):
        super().__init__()
        g = 1 if in_channels < num_groups else num_groups  # buildingblocks.py:66-68
        assert in_channels % g == 0, (
            f"Expected number of channels in input to be divisible by num_groups. "
            f"num_channels={in_channels}, num_groups={g}")
        self.groupnorm = nn.GroupNorm(num_groups=g, num_channels=in_channels)
        self.conv = nn.Conv3d(in_channels, out_channels, 3, padding=1, bias=False)
        self._groups = g
        self._first = first_layer
        # set by AbstractUNet: True when every consumer of this layer's output is another SingleConv (directly or
        # through max-pool / upsample+concat), whose backward already masks its dx by (x > 0)
        self._dy_premasked = False

    def forward(self, x, dy_premasked=None, dy_blocked=False, dx_blocked=False, pool=False, dy_lazy=False,
                dx_lazy=False, dgrad_terms=0):  # x NDHWC
        """dy_premasked overrides the static promise for this call (the fused keypoint head masks its feature gradient);
        dy_blocked / dx_blocked: see DoubleConv.forward; pool: return maxpool2 of the output (B.conv_pool_ok)."""
        return B.single_conv_gcr(x, self.groupnorm.weight, self.groupnorm.bias, self.conv.weight, self._groups,
                                 x_from_relu=not self._first,
                                 dy_premasked=self._dy_premasked if dy_premasked is None else dy_premasked,
                                 dy_blocked=dy_blocked, dx_blocked=dx_blocked, pool=pool, dy_lazy=dy_lazy, dx_lazy=dx_lazy,
                                 dgrad_terms=dgrad_terms)


class DoubleConv(nn.Module):
    def __init__(self, in_channels, out_channels, encoder, num_groups=8, first_layer=False):
        super().__init__()
        if encoder:
            c1_out = max(out_channels // 2, in_channels)
            c1 = (in_channels, c1_out)
            c2 = (c1_out, out_channels)
        else:
            c1 = (in_channels, out_channels)
            c2 = (out_channels, out_channels)
        self.SingleConv1 = SingleConv(*c1, num_groups=num_groups, first_layer=first_layer)
        self.SingleConv2 = SingleConv(*c2, num_groups=num_groups)
        self.SingleConv1._dy_premasked = True     # its only consumer, SingleConv2, masks the gradient it returns by (x > 0)

    def forward(self, x, out_premasked=None, out_dy_blocked=False, out_pool=False):
        # out_pool: the block's output goes ONLY to a max-pool, which the second convolution then applies in its epilogue
        # (the block returns the POOLED tensor; out_dy_blocked then refers to the scattered gradient inside that operator)
        # out_dy_blocked: the block's output goes ONLY to a max-pool called with blocked_grad=True
        # the hidden activation has exactly one consumer, so its gradient can travel in the layout the first conv's
        # gradient kernels read fastest (channel-blocked, backbone_ops.grad_blocked_ok) -- an internal hand-off
        n, d, h, w, cin = x.shape
        blk = (torch.is_grad_enabled() and self.SingleConv1._dy_premasked
               and B.grad_blocked_ok(n, d, h, w, cin, self.SingleConv1.conv.out_channels))
        # first encoder block on an image: GroupNorm's backward of the second convolution is applied inside the first
        # layer's correlation kernel instead of being written out (B.lazy_first_layer_ok)
        lazy = (self.SingleConv1._first and self.SingleConv1._dy_premasked and not blk
                and B.lazy_first_layer_ok(x, self.SingleConv1.conv.out_channels))
        return self.SingleConv2(self.SingleConv1(x, dy_blocked=blk, dy_lazy=lazy), out_premasked,
                                dy_blocked=out_dy_blocked, dx_blocked=blk, pool=out_pool, dx_lazy=lazy,
                                dgrad_terms=B.first_block_dgrad_terms() if self.SingleConv1._first else 0)


class Encoder(nn.Module):
    def __init__(self, in_channels, out_channels, apply_pooling=True, num_groups=8, first_layer=False):
        super().__init__()
        self.apply_pooling = apply_pooling
        self.basic_module = DoubleConv(in_channels, out_channels, True, num_groups, first_layer)

    def forward(self, x):
        if self.apply_pooling:
            x = B.maxpool2(x)
        return self.basic_module(x)


class Decoder(nn.Module):
    def __init__(self, in_channels, out_channels, num_groups=8):
        super().__init__()
        self.basic_module = DoubleConv(in_channels, out_channels, False, num_groups)

    def forward(self, encoder_features, x, lazy_skip_grad=False, out_premasked=None):
        dc = self.basic_module
        c1 = dc.SingleConv1
        if B.upcat_conv_ok(encoder_features, x, c1.conv.out_channels):
            # interpolate + cat + the first SingleConv as one operator: no concatenated tensor, the upsampled channels
            # are convolved (forward) and differentiated (backward) at low resolution
            # lazy_skip_grad: the skip tensor is pool_fork's second output, so its gradient goes to that node's backward
            # only -- which then applies this operator's GroupNorm backward for the skip half itself (one pass less)
            # blk: h has one consumer, so its gradient travels channel-blocked (whole cache lines for the four kernels
            # of the fused operator's backward that read it)
            blk = B.upcat_blocked_ok(encoder_features, x, c1.conv.out_channels)
            h = B.upcat_conv_gcr(encoder_features, x, c1.groupnorm.weight, c1.groupnorm.bias, c1.conv.weight,
                                 c1._groups, dy_premasked=True,     # its only consumer, SingleConv2, masks its dx
                                 dskip_lazy=bool(lazy_skip_grad) and B.lazy_skip_ok(encoder_features), dy_blocked=blk)
            return dc.SingleConv2(h, out_premasked, dx_blocked=blk)
        return dc(B.upcat(encoder_features, x, lazy_skip_grad), out_premasked)


class AbstractUNet(nn.Module):
    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8,
                 num_levels=4, is_segmentation=True, conv_padding=1, num_truncated_layers=0, use_checkpoint=False, **kwargs):
        super().__init__()
        # keymorph/unet3d/model.py:113, :119-144: every encoder / decoder block under torch.utils.checkpoint (non-reentrant):
        # a block's saved activations are dropped after its forward and recomputed, block by block, during the backward.
        # The kernels are deterministic, so gradients are bit-identical with and without it (test_use_checkpoint_*).
        self.use_checkpoint = bool(use_checkpoint)
        if layer_order != "gcr" or conv_padding != 1:
            raise NotImplementedError("keymorph_amd implements the 'gcr', padding=1 U-Net of scripts/run.py")
        if isinstance(f_maps, int):
            f_maps = number_of_features_per_level(f_maps, num_levels=num_levels)
        assert isinstance(f_maps, (list, tuple)) and len(f_maps) > 1, "Required at least 2 levels in the U-Net"
        self.f_maps = list(f_maps)
        self.encoders = nn.ModuleList(
            Encoder(in_channels if i == 0 else f_maps[i - 1], f, apply_pooling=i > 0, num_groups=num_groups,
                    first_layer=i == 0)
            for i, f in enumerate(f_maps))
        rf = list(reversed(f_maps))
        decs = [Decoder(rf[i] + rf[i + 1], rf[i + 1], num_groups) for i in range(len(rf) - 1)]
        if num_truncated_layers > 0:
            decs = decs[:-num_truncated_layers]
        self.decoders = nn.ModuleList(decs)
        self.final_conv = nn.Conv3d(f_maps[num_truncated_layers], out_channels, 1)
        # every SingleConv output feeds only SingleConvs, except the one in front of the 1x1x1 head
        convs = [m for m in self.modules() if isinstance(m, SingleConv)]
        last_block = (self.decoders[-1] if len(self.decoders) else self.encoders[-1]).basic_module
        for m in convs:
            m._dy_premasked = m is not last_block.SingleConv2
        if is_segmentation:
            self.final_activation = nn.Sigmoid() if final_sigmoid else nn.Softmax(dim=1)
        else:
            self.final_activation = None

    def features(self, x, head_masks_gradient=False):
        """(N,1,D,H,W) image -> NDHWC feature map in front of the final 1x1x1 conv.
        head_masks_gradient: the caller's consumer (the fused keypoint head with feat_from_relu=True) returns the
        feature gradient already multiplied by (feat > 0), so the last block skips its ReLU-backward mask too."""
        last_pm = True if head_masks_gradient else None
        x = B.to_ndhwc(x)
        feats = []
        # encoder outputs that feed BOTH the next level's pooling and a decoder's skip connection go through
        # pool_fork: one fused backward pass sums their two gradients
        L, nd = len(self.encoders), len(self.decoders)
        forked = {L - 2 - j for j in range(nd)}
        pooled = None
        prev_blk = False
        for i, enc in enumerate(self.encoders):
            opm = last_pm if (nd == 0 and i == L - 1) else None
            # an encoder output that feeds ONLY the next level's pooling hands its gradient over channel-blocked
            nxt_pools = i + 1 < L and self.encoders[i + 1].apply_pooling and i not in forked
            c2 = enc.basic_module.SingleConv2
            blk_out = (nxt_pools and torch.is_grad_enabled() and c2._dy_premasked and opm is None
                       and c2.conv.out_channels % 8 == 0)
            if i == 0 or not enc.apply_pooling:
                xin = x
            else:
                xin = pooled if pooled is not None else B.maxpool2(x, blocked_grad=prev_blk)
            if blk_out:
                n_, d_, h_, w_ = xin.shape[0], xin.shape[1], xin.shape[2], xin.shape[3]
                blk_out = (d_ % 2 == 0 and h_ % 2 == 0 and w_ % 2 == 0 and
                           B.grad_blocked_ok(n_, d_, h_, w_, c2.conv.in_channels, c2.conv.out_channels))
            # ... and, when the shapes allow, never exists at full resolution: the second convolution pools in its epilogue
            fuse_pool = (nxt_pools and opm is None and c2._dy_premasked
                         and B.conv_pool_ok(xin.shape[0], xin.shape[1], xin.shape[2], xin.shape[3], c2.conv.in_channels,
                                            c2.conv.out_channels))
            x = self._block(enc.basic_module, xin, opm, out_dy_blocked=blk_out, out_pool=fuse_pool)
            prev_blk = blk_out
            pooled = x if fuse_pool else None
            if fuse_pool:
                feats.insert(0, (None, False))       # (not a skip source: the truncated / pooled-only level)
                continue
            if i in forked and i + 1 < L and self.encoders[i + 1].apply_pooling:
                pooled, x = B.pool_fork(x)
                feats.insert(0, (x, True))
            else:
                feats.insert(0, (x, False))
        x = feats[0][0]
        for j, (dec, (skip, lazy)) in enumerate(zip(self.decoders, feats[1:])):
            x = self._block(dec, skip, x, lazy, last_pm if j == nd - 1 else None)
        return x

    def _block(self, fn, *args, **kwargs):
        """one encoder / decoder block, checkpointed when use_checkpoint (and a backward can follow)"""
        if self.use_checkpoint and torch.is_grad_enabled():
            from torch.utils import checkpoint
            return checkpoint.checkpoint(lambda *a: fn(*a, **kwargs), *args, use_reentrant=False)
        return fn(*args, **kwargs)

    def keypoints_ij(self, x):
        """CenterOfMass3d('ij')(forward(x)) with the 1x1x1 head, ReLU and the center of mass fused (no heat-map).
        Only defined for the regression configuration (no final activation), which is what KeyMorph builds."""
        assert self.final_activation is None or self.training
        fused = self.final_conv.in_channels <= B.HEAD_FUSED_MAX_CIN
        feat = self.features(x, head_masks_gradient=fused)
        if not fused:
            from .. import ops
            return ops.com3d(B.pointwise(feat, self.final_conv.weight, self.final_conv.bias))
        return B.head_com(feat, self.final_conv.weight, self.final_conv.bias, feat_from_relu=True)

    def keypoints_and_power(self, x):
        """Differentiable (keypoints (N,K,3) ij, power (N,K) = sum relu(h)) for training with
        weight_keypoints='power' (keymorph/model.py:96-109, :183-191): one fused head pass, one fused backward."""
        assert self.final_activation is None or self.training
        if self.final_conv.in_channels > B.HEAD_FUSED_MAX_CIN:
            raise NotImplementedError("keypoint weighting needs the fused head (final conv with <= 64 input channels)")
        feat = self.features(x, head_masks_gradient=True)
        return B.head_com_power(feat, self.final_conv.weight, self.final_conv.bias, feat_from_relu=True)

    def keypoints_and_moments(self, x):
        """Inference only: (keypoints (N,K,3) ij, sum relu(h) (N,K), sum relu(h)^2 (N,K), voxels per channel) for
        keypoint weighting (keymorph/model.py:75-109), from the fused head -- no heat-map."""
        assert self.final_activation is None or self.training
        feat = self.features(x)
        if feat.shape[-1] > B.HEAD_FUSED_MAX_CIN:
            raise NotImplementedError("keypoint weighting needs the fused head (final conv with <= 64 input channels)")
        pts, power, sq = B.head_moments(feat, self.final_conv.weight, self.final_conv.bias)
        return pts, power, sq, feat.shape[1] * feat.shape[2] * feat.shape[3]

    def forward(self, x):
        y = B.pointwise(self.features(x), self.final_conv.weight, self.final_conv.bias)
        if not self.training and self.final_activation is not None:
            y = self.final_activation(y)
        return y


class UNet3D(AbstractUNet):
    """keymorph/unet3d/model.py:154-189"""

    def __init__(self, in_channels, out_channels, final_sigmoid=True, f_maps=64, layer_order="gcr", num_groups=8,
                 num_levels=4, is_segmentation=True, conv_padding=1, use_checkpoint=False, **kwargs):
        super().__init__(in_channels, out_channels, final_sigmoid, f_maps, layer_order, num_groups, num_levels,
                         is_segmentation, conv_padding, 0, use_checkpoint=use_checkpoint)


class TruncatedUNet3D(AbstractUNet):
    """keymorph/unet3d/model.py:394-430"""

    def __init__(self, in_channels, out_channels, num_truncated_layers, final_sigmoid=True, f_maps=64,
                 layer_order="gcr", num_groups=8, num_levels=4, is_segmentation=True, conv_padding=1,
                 use_checkpoint=False, **kwargs):
        super().__init__(in_channels, out_channels, final_sigmoid, f_maps, layer_order, num_groups, num_levels,
                         is_segmentation, conv_padding, num_truncated_layers, use_checkpoint=use_checkpoint)


# Names the reference's scripts import beside UNet3D / TruncatedUNet3D (scripts/run.py:13, scripts/register.py:11) and that
# are not on the 3-D registration path: they exist so that those import lines succeed, and raise when constructed.
from .._absent import absent_class as _absent_class, absent_function as _absent_function   # noqa: E402

UNet2D = _absent_class("UNet2D", "keymorph/unet3d/model.py:266", nn.Module)
ResidualUNet3D = _absent_class("ResidualUNet3D", "keymorph/unet3d/model.py:192", nn.Module)
ResidualUNetSE3D = _absent_class("ResidualUNetSE3D", "keymorph/unet3d/model.py:228", nn.Module)
AbstractTruncatedUNet = AbstractUNet          # keymorph/unet3d/model.py:307: here one class takes `num_truncated_layers`
get_model = _absent_function("get_model", "keymorph/unet3d/model.py:300")
