"""``LargerAE`` (reference networks/acai_vanilla_modified.py:22-105): depth//2 stem in the encoder and a 1x1 conv
(+BatchNorm) in front of the decoder; executed on the HIP engine like ``VanillaACAI``."""
from .acai_vanilla import Decoder as _Decoder
from .acai_vanilla import Encoder as _Encoder
from .acai_vanilla import HipAE, Initializer, num_scales  # noqa: F401


def Encoder(scales, depth, latent, colors, n_res_block=None, use_batchnorm=False):
    return _Encoder(scales, depth, latent, colors, n_res_block, use_batchnorm, stem=depth // 2)


def Decoder(scales, depth, latent, colors, n_res_block=None, use_upsample=True, use_batchnorm=False, use_sigmoid=False):
    return _Decoder(scales, depth, latent, colors, n_res_block, use_upsample, use_batchnorm, use_sigmoid, stem_1x1=True)


def create_decoder(args):
    return Decoder(num_scales(args), args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                   use_batchnorm=args["use_batchnorm"], use_sigmoid=args["use_sigmoid"]).to(args["device"])


class LargerAE(HipAE):
    fixed_depth = 3

    def __init__(self, args):
        super().__init__()
        scales = num_scales(args)
        self._fill_defaults(args)
        self.enc = Encoder(scales, args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                           use_batchnorm=args["use_batchnorm"]).to(args["device"])
        self.dec = Decoder(scales, args["depth"], args["latent"], args["colors"], n_res_block=args["n_res_block"],
                           use_batchnorm=args["use_batchnorm"], use_sigmoid=args["use_sigmoid"]).to(args["device"])
