"""GMF (openrec/tf2/recommenders/gmf.py:5-41): logit = Dense(1, no bias)(u * i) + b_i,
binary cross-entropy with logits (mean), l2 over u, i and the Dense kernel."""
from ._base import Recommender, _ids
from ..modules import MLP
from ... import runtime as rt


class GMF(Recommender):
    _score_kind = "gmf"

    def __init__(self, dim_user_embed, dim_item_embed, total_users, total_items, ctx=None):
        self._build_tables(dim_user_embed, dim_item_embed, total_users, total_items, ctx)
        self.mlp = MLP(units_list=[1], use_bias=False).build(dim_user_embed, ctx)

    @property
    def trainable_variables(self):
        return super().trainable_variables + self.mlp.trainable_variables

    def __call__(self, user_id, item_id, label):
        U, V, b = self._tables()
        w = self.mlp.layers[0].kernel
        uid, iid, lab = _ids(user_id), _ids(item_id), _ids(label)

        def run_forward():
            return rt.pointwise_loss("gmf", U, V, b, w, uid, iid, lab)

        def run_train(optimizer, no_l2):
            loss, l2 = rt.pointwise_step("gmf", optimizer, U, V, b, w, uid, iid, lab, K=1, no_l2=no_l2)
            return float(loss[0]), float(l2[0])

        return self._record(run_forward, run_train)

    call = __call__

    def inference(self, user_id):
        """gmf.py:36-41:  sum_d w_d u_d V_d + b."""
        U, V, b = self._tables()
        return rt.score_all_items("gmf", U, V, b, _ids(user_id), w=self.mlp.layers[0].kernel)
