"""The recommenders of `openrec.tf2.recommenders` (same class names, constructor arguments and call signatures),
each running its train step as a fused device call: see the file of the same name."""
from . import bpr as _bpr, dlrm as _dlrm, gmf as _gmf, ucml as _ucml, wrmf as _wrmf

BPR, UCML, GMF, WRMF, DLRM = _bpr.BPR, _ucml.UCML, _gmf.GMF, _wrmf.WRMF, _dlrm.DLRM

__all__ = ["BPR", "UCML", "GMF", "WRMF", "DLRM"]
