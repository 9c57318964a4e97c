"""BPR (openrec/tf2/recommenders/bpr.py:5-43): same constructor, attributes,
call signature and return value `(loss, l2_loss)`; the five gathers, the
pairwise log loss, l2_loss and (under a tape) the gradients + optimizer update
run as one fused HIP kernel."""
from ._base import PairwiseRecommender, _ids
from ..modules import PairwiseLogLoss
from ... import runtime as rt


class BPR(PairwiseRecommender):
    _model = "bpr"

    def __init__(self, dim_user_embed, dim_item_embed, total_users, total_items, ctx=None):
        self._build_tables(dim_user_embed, dim_item_embed, total_users, total_items, ctx)
        self.pairwise_log_loss = PairwiseLogLoss()

    def inference(self, user_id):
        """bpr.py:39-43:  U[user_id] @ V^T + b  -> [B, total_items]."""
        U, V, b = self._tables()
        return rt.score_all_items("dot", U, V, b, _ids(user_id), device=True)
