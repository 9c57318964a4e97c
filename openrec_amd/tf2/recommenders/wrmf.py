"""WRMF (openrec/tf2/recommenders/wrmf.py:5-40): prediction = u . i + b_i,
confidence-weighted squared error summed over the batch
(modules/pointwise_mse_loss.py:18-31)."""
from ._base import PointwiseRecommender, _ids
from ..modules import PointwiseMSELoss
from ... import runtime as rt


class WRMF(PointwiseRecommender):

    def __init__(self, dim_user_embed, dim_item_embed, total_users, total_items, a=1.0, b=1.0, ctx=None):
        self._build_tables(dim_user_embed, dim_item_embed, total_users, total_items, ctx)
        self.pointwise_mse_loss = PointwiseMSELoss(a=a, b=b)
        self._a, self._b = a, b

    def _point_args(self):
        return "wrmf", None, dict(a=self._a, b_w=self._b)

    def inference(self, user_id):
        """wrmf.py:36-40:  U[user_id] @ V^T + b."""
        U, V, b = self._tables()
        return rt.score_all_items("dot", U, V, b, _ids(user_id), device=True)
