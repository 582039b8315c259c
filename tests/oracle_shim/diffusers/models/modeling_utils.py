import torch.nn as nn


class ModelMixin(nn.Module):
    @property
    def dtype(self):
        return next(self.parameters()).dtype
