"""WRMF (openrec/tf2/recommenders/wrmf.py:5-40): prediction = u . i + b_i,
confidence-weighted squared error summed over the batch
(modules/pointwise_mse_loss.py:18-31)."""
from ._base import Recommender, _ids
from ..modules import PointwiseMSELoss
from ... import runtime as rt


class WRMF(Recommender):

    def __init__(self, dim_user_embed, dim_item_embed, total_users, total_items, a=1.0, b=1.0, ctx=None):
        self._build_tables(dim_user_embed, dim_item_embed, total_users, total_items, ctx)
        self.pointwise_mse_loss = PointwiseMSELoss(a=a, b=b)
        self._a, self._b = a, b

    def __call__(self, user_id, item_id, label):
        U, V, b = self._tables()
        uid, iid, lab = _ids(user_id), _ids(item_id), _ids(label)

        def run_forward():
            return rt.pointwise_loss("wrmf", U, V, b, None, uid, iid, lab, a=self._a, b_w=self._b)

        def run_train(optimizer, no_l2):
            loss, l2 = rt.pointwise_step("wrmf", optimizer, U, V, b, None, uid, iid, lab, K=1,
                                         a=self._a, b_w=self._b, no_l2=no_l2)
            return float(loss[0]), float(l2[0])

        return self._record(run_forward, run_train)

    call = __call__

    def inference(self, user_id):
        """wrmf.py:36-40:  U[user_id] @ V^T + b."""
        U, V, b = self._tables()
        return rt.score_all_items("dot", U, V, b, _ids(user_id))
