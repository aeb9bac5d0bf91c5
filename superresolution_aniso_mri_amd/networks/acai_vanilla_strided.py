"""``VanillaACAIStrided`` (reference networks/acai_vanilla_strided.py:9-55): the AvgPool2d(2) of every encoder scale is
replaced by ``Conv2d(k, k, 2, stride=2)``; on the HIP engine that layer runs as space-to-depth + a 1x1 MFMA convolution."""
from .acai_vanilla import Decoder, HipAE, Initializer, num_scales  # noqa: F401
from .acai_vanilla import Encoder as _Encoder


def Encoder(scales, depth, latent, colors, n_res_block=None, use_batchnorm=False):
    return _Encoder(scales, depth, latent, colors, n_res_block, use_batchnorm, downsample="strided")


class VanillaACAIStrided(HipAE):

    def __init__(self, args):
        super().__init__()
        scales = num_scales(args)
        self._fill_defaults(args)
        self.enc = Encoder(scales, args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                           use_batchnorm=args["use_batchnorm"]).to(args["device"])
        self.dec = Decoder(scales, args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                           use_batchnorm=args["use_batchnorm"], use_sigmoid=args["use_sigmoid"]).to(args["device"])
