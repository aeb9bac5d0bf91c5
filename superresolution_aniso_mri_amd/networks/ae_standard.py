"""MI355X-native building blocks of the reference's older block-style auto-encoder (networks/ae_standard.py:34-80):
``BasicEncoderBlock`` (conv Cin->Cin, LeakyReLU, conv Cin->Cout, LeakyReLU, [BatchNorm2d], [AvgPool2d(2)]) and
``BasicDecoderBlock`` (conv, LeakyReLU, conv, LeakyReLU, [bilinear x2 Upsample, align_corners=False]) with the reference's
constructor arguments and parameter names (``conv2d_1``, ``conv2d_2``, ``batchnorm``); ``forward`` runs on the HIP engine
(stand-alone pooling / bilinear steps: csrc/resample.hip).

The reference's ``AE`` / ``AEAdv`` / ``DiscriminatorSpatial`` classes need ``networks.model_configs``, which does not exist
in the reference, and have no caller; the blocks are the part of that file on the hot path.
``BlockStack`` chains blocks into one engine pass (what ``nn.Sequential(self.encoder, block)`` does in the reference).
Channel counts must be multiples of 4 (MFMA convolution path)."""
import torch.nn as nn

from .. import engine


def weights_init(m):
    """networks/ae_standard.py:6-10"""
    if m.__class__.__name__.find("Conv2d") != -1:
        nn.init.kaiming_normal_(m.weight)
        m.bias.data.zero_()


class _HipBlock(nn.Module):
    def primitive_layers(self):
        raise NotImplementedError

    def _runner(self):
        r = self.__dict__.get("_aesr_runner")
        if r is None:
            r = engine.SequentialRunner(nn.Sequential(*self.primitive_layers()))
            self.__dict__["_aesr_runner"] = r
        return r

    def forward(self, x):
        out = engine.run_pass(self._runner(), engine.to_nhwc(x), train=self.training)
        return engine.to_nchw_view(out)


class BasicEncoderBlock(_HipBlock):
    """networks/ae_standard.py:34-57"""

    def __init__(self, channels_in, channels_out, kernel=3, padding=1, downsample=True, use_batchnorm=False, dropout_perc=0.):
        super().__init__()
        if dropout_perc != 0:
            raise NotImplementedError("Dropout2d is not part of the ae_combined path (the reference never enables it)")
        self.conv2d_1 = nn.Conv2d(channels_in, channels_in, kernel, stride=1, dilation=1, padding=padding)
        self.non_linear = nn.LeakyReLU()
        self.conv2d_2 = nn.Conv2d(channels_in, channels_out, kernel, stride=1, dilation=1, padding=padding)
        self.batchnorm = nn.BatchNorm2d(channels_out)
        self.max_pool = nn.AvgPool2d(2, stride=None, padding=0)
        self.downsample, self.use_batchnorm, self.dropout_perc = downsample, use_batchnorm, dropout_perc

    def primitive_layers(self):
        layers = [self.conv2d_1, self.non_linear, self.conv2d_2, nn.LeakyReLU()]
        if self.use_batchnorm:
            layers.append(self.batchnorm)
        if self.downsample:
            layers.append(self.max_pool)
        return layers


class BasicDecoderBlock(_HipBlock):
    """networks/ae_standard.py:60-80 (the BatchNorm2d it constructs is never applied by the reference's forward)"""

    def __init__(self, channels_in, channels_out, kernel=3, padding=1, do_upsample=True, dropout_perc=0.):
        super().__init__()
        if dropout_perc != 0:
            raise NotImplementedError("Dropout2d is not part of the ae_combined path (the reference never enables it)")
        self.conv2d_1 = nn.Conv2d(channels_in, channels_in, kernel, stride=1, dilation=1, padding=padding)
        self.non_linear = nn.LeakyReLU()
        self.conv2d_2 = nn.Conv2d(channels_in, channels_out, kernel, stride=1, dilation=1, padding=padding)
        self.batchnorm = nn.BatchNorm2d(channels_out)
        self.upsample = nn.Upsample(scale_factor=2, mode="bilinear", align_corners=False)
        self.do_upsample, self.dropout_perc = do_upsample, dropout_perc

    def primitive_layers(self):
        layers = [self.conv2d_1, self.non_linear, self.conv2d_2, nn.LeakyReLU()]
        if self.do_upsample:
            layers.append(self.upsample)
        return layers


class BlockStack(_HipBlock):
    """Several blocks (and plain Conv2d / LeakyReLU / Sigmoid modules) as ONE engine pass."""

    def __init__(self, *blocks):
        super().__init__()
        self.blocks = nn.ModuleList(blocks)

    def primitive_layers(self):
        layers = []
        for b in self.blocks:
            layers += b.primitive_layers() if isinstance(b, _HipBlock) else [b]
        return layers
