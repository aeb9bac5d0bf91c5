"""``PerceptualLoss`` (reference lpips/perceptual.py:6-33): callable ``(pred, target, normalize=False) -> [N,1,1,1]``."""
import torch

from .dist_model import DistModel


class PerceptualLoss(torch.nn.Module):
    def __init__(self, model="net-lin", net="alex", colorspace="rgb", spatial=False, use_gpu=True, gpu_ids=[0], vgg_weights=None,
                 device=None):
        super(PerceptualLoss, self).__init__()
        self.use_gpu, self.spatial, self.gpu_ids = use_gpu, spatial, gpu_ids
        self.model = DistModel()
        self.model.initialize(model=model, net=net, use_gpu=use_gpu, colorspace=colorspace, spatial=spatial, gpu_ids=gpu_ids,
                              vgg_weights=vgg_weights, device=device)

    def forward(self, pred, target, normalize=False):
        """normalize=True: images in [0,1] are mapped to [-1,1] (folded into the first convolution's loader).
        Note the reference's argument order: the distance network is called as (target, pred)."""
        return self.model.forward(target, pred, _affine=(2.0, -1.0) if normalize else (1.0, 0.0))
