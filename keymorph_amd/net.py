"""ConvNet backbone with the reference's surface and state_dict keys (keymorph/net.py:7-36,
keymorph/layers.py:137-187): 9 x [Conv3d(k3,p1,bias) -> norm -> ReLU (-> MaxPool3d(2))],
channels 32,64,64,128,128,256,256,512,out_dim; pooling after blocks 2,4,6,8."""
import torch.nn as nn

from . import backbone_ops as B

h_dims = [32, 64, 64, 128, 128, 256, 256, 512]


class ConvBlock(nn.Module):
    def __init__(self, in_channels, out_channels, stride, norm_type, down_sample=True, dim=3):
        super().__init__()
        if dim != 3:
            raise NotImplementedError("keymorph_amd implements the 3-D registration path")
        if stride != 1:
            raise NotImplementedError("stride 1 only (the only value keymorph/net.py uses)")
        self.norm_type = norm_type
        self.down_sample = down_sample
        if norm_type == "none":
            self.norm = None
        elif norm_type == "instance":
            self.norm = nn.InstanceNorm3d(out_channels)      # affine=False: no parameters, holder only
        elif norm_type == "group":
            self.norm = nn.GroupNorm(num_groups=8, num_channels=out_channels)
        elif norm_type == "batch":
            self.norm = nn.BatchNorm3d(out_channels)         # parameters + running statistics, same state_dict keys
        else:
            raise NotImplementedError(norm_type)
        self.conv = nn.Conv3d(in_channels, out_channels, kernel_size=3, stride=1, padding=1)  # parameter holder
        self._cout = out_channels

    def forward(self, x):  # NDHWC
        if self.norm_type == "none":
            out = B.conv_block(x, self.conv.weight, self.conv.bias, None, None, 0)
        elif self.norm_type == "instance":
            out = B.conv_block(x, self.conv.weight, self.conv.bias, None, None, self._cout)
        elif self.norm_type == "batch":
            out = B.conv_block_batchnorm(x, self.conv.weight, self.conv.bias, self.norm, self.training)
        else:
            out = B.conv_block(x, self.conv.weight, self.conv.bias, self.norm.weight, self.norm.bias, 8)
        if self.down_sample:
            out = B.maxpool2(out)
        return out


class ConvNet(nn.Module):
    def __init__(self, dim, input_ch, out_dim, norm_type):
        super().__init__()
        self.dim = dim
        chans = [input_ch] + h_dims + [out_dim]
        for b in range(1, 10):
            setattr(self, f"block{b}", ConvBlock(chans[b - 1], chans[b], 1, norm_type, b in (2, 4, 6, 8), dim))

    def forward(self, x):
        out = B.to_ndhwc(x)
        blocks = [getattr(self, f"block{b}") for b in range(1, 10)]
        if B.convnet_lazy_ok(out, blocks[0].norm_type):
            # InstanceNorm + ReLU + MaxPool applied inside the NEXT convolution's loader: nothing but the raw convolution
            # outputs is stored (backbone_ops.convnet_instance_lazy)
            out = B.convnet_instance_lazy(out, [(m.conv.weight, m.conv.bias, m.down_sample) for m in blocks])
        else:
            for m in blocks:
                out = m(out)
        return B.to_ncdhw(out)
