"""The module classes a script imports from `openrec.tf2.modules`, re-implemented on
HBM-resident tables: every class lives in the file of the same name next to this one."""
from . import latent_factor as _lf
from . import multi_layer_perceptron as _mlp
from . import pairwise_log_loss as _pll
from . import pointwise_mse_loss as _pml
from . import second_order_feature_interaction as _sofi

LatentFactor = _lf.LatentFactor
MLP = _mlp.MLP
PairwiseLogLoss = _pll.PairwiseLogLoss
PointwiseMSELoss = _pml.PointwiseMSELoss
SecondOrderFeatureInteraction = _sofi.SecondOrderFeatureInteraction

__all__ = ["LatentFactor", "MLP", "PairwiseLogLoss", "PointwiseMSELoss", "SecondOrderFeatureInteraction"]
