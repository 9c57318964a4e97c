"""GMF (openrec/tf2/recommenders/gmf.py:5-41): logit = Dense(1, no bias)(u * i) + b_i,
binary cross-entropy with logits (mean), l2 over u, i and the Dense kernel."""
from ._base import PointwiseRecommender, _ids
from ..modules import MLP
from ... import runtime as rt


class GMF(PointwiseRecommender):
    _score_kind = "gmf"

    def __init__(self, dim_user_embed, dim_item_embed, total_users, total_items, ctx=None):
        self._build_tables(dim_user_embed, dim_item_embed, total_users, total_items, ctx)
        self.mlp = MLP(units_list=[1], use_bias=False).build(dim_user_embed, ctx)
        self.mlp.layers[0].kernel.pre_access = self.flush

    @property
    def trainable_variables(self):
        return super().trainable_variables + self.mlp.trainable_variables

    def _point_args(self):
        return "gmf", self.mlp.layers[0].kernel, {}

    def inference(self, user_id):
        """gmf.py:36-41:  sum_d w_d u_d V_d + b."""
        U, V, b = self._tables()
        return rt.score_all_items("gmf", U, V, b, _ids(user_id), w=self.mlp.layers[0].kernel, device=True)
