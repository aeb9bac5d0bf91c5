"""``DistModel`` wrapper (reference lpips/dist_model.py:19-108) for model='net-lin', net='vgg' in test mode."""
import warnings

import torch

from . import networks_basic as networks
from .synthetic_vgg import synthetic_vgg16_state


class DistModel(object):
    def name(self):
        return self.model_name

    def initialize(self, model="net-lin", net="vgg", colorspace="Lab", pnet_rand=False, pnet_tune=False, model_path=None,
                   use_gpu=True, printNet=False, spatial=False, is_train=False, lr=.0001, beta1=0.5, version="0.1",
                   gpu_ids=[0], vgg_weights=None, device=None):
        if model != "net-lin" or net not in ("vgg", "vgg16") or is_train:
            raise ValueError("Model [%s/%s] not available in this build (net-lin / vgg, test mode)" % (model, net))
        self.model, self.net_name, self.spatial, self.gpu_ids = model, net, spatial, gpu_ids
        self.model_name = "%s [%s]" % (model, net)
        if vgg_weights in (None, "synthetic-hash"):
            warnings.warn("LPIPS: no local vgg16 state_dict given (--vgg_weights); using the deterministic SYNTHETIC backbone. "
                          "Distances are not the published LPIPS metric.")
            sd = synthetic_vgg16_state()
        else:
            sd = torch.load(vgg_weights, map_location="cpu")
        self.net = networks.PNetLin(pnet_rand=pnet_rand, pnet_tune=pnet_tune, pnet_type=net, use_dropout=True, spatial=spatial,
                                    version=version, lpips=True, vgg_state_dict=sd)
        networks.load_lin_weights(self.net, model_path)
        self.net.eval()
        for p in self.net.parameters():
            p.requires_grad = False            # the loss network is a constant of the optimisation
        dev = device if device is not None else ("cuda:%d" % gpu_ids[0] if use_gpu else "cpu")
        self.net.to(dev)
        self.parameters = list(self.net.parameters())

    def forward(self, in0, in1, retPerLayer=False, **kw):
        return self.net.forward(in0, in1, retPerLayer=retPerLayer, **kw)
