from .bpr import BPR
from .ucml import UCML
from .gmf import GMF
from .wrmf import WRMF
from .dlrm import DLRM
