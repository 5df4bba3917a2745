"""Mirror of model/pretrained/core.py:8-20 (BaseModel): optimizer/scheduler recipe only."""
from abc import ABC, abstractmethod

import torch
import torch.nn as nn
from torch.optim.lr_scheduler import CosineAnnealingLR, LinearLR, SequentialLR


class BaseModel(nn.Module, ABC):
    def __init__(self):
        super().__init__()

    @abstractmethod
    def shared_eval(self, batch, optimizer, scheduler, mode):
        ...

    def configure_optimizers(self, lr=1e-3):
        """core.py:15-20: AdamW(wd 1e-2); 1000-iteration linear warm-up then cosine (T_max = 400-1000
        as written in the reference, eta_min 1e-6)."""
        opt = torch.optim.AdamW(self.parameters(), lr=lr, weight_decay=1e-2)
        warm = LinearLR(opt, start_factor=0.1, total_iters=1000)
        cos = CosineAnnealingLR(opt, T_max=400 - 1000, eta_min=1e-6)
        return opt, SequentialLR(opt, schedulers=[warm, cos], milestones=[1000])


BaseModel.__module__ = "model.pretrained.core"
