"""ModelRepresentation -- mirrors the reference's model/wrapper.py:7-51.

In eval the reference wrapper is a pass-through (`return self.model(x)`, model/wrapper.py:50-51);
the "rep" head only exists for U2PL training and is out of scope, so requesting it raises.
"""
from torch import nn


class ModelRepresentation(nn.Module):
    def __init__(self, model, rep=None, rep_forward=None, x_tmp_transform=None, *args, **kwargs):
        super().__init__(*args, **kwargs)
        self.model = model
        self.rep = rep
        self.x_tmp_transform = x_tmp_transform

    def forward(self, x):
        if self.training:
            raise NotImplementedError("ModelRepresentation(HIP): the 'rep' output exists only for training (model/wrapper.py:34-49)")
        return self.model(x)
