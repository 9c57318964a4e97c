"""UCML (openrec/tf2/recommenders/ucml.py:5-53): squared-L2 scores, hinge-sum
loss with `margin`, `censor_vec` after the step."""
from ._base import PairwiseRecommender, _ids
from ... import runtime as rt


class UCML(PairwiseRecommender):
    _model = "ucml"
    _score_kind = "l2"

    def __init__(self, dim_user_embed, dim_item_embed, total_users, total_items, margin=0.5, ctx=None):
        self._build_tables(dim_user_embed, dim_item_embed, total_users, total_items, ctx)
        self.margin = margin

    def censor_vec(self, user_id, p_item_id, n_item_id):
        """ucml.py:44-48: users, then positive items, then negative items (sequential).  Right after a queued
        train step on the same ids it becomes that step's in-kernel censor (ORX_CENSOR) instead of three passes."""
        if self._queue.mark_censor((_ids(user_id), _ids(p_item_id), _ids(n_item_id))):
            return tuple(self.trainable_variables[:2]) + (self.trainable_variables[1],)
        return (self.user_latent_factor.censor(_ids(user_id)),
                self.item_latent_factor.censor(_ids(p_item_id)),
                self.item_latent_factor.censor(_ids(n_item_id)))

    def inference(self, user_id):
        """ucml.py:50-53:  -||U[user] - V||^2 + b  -> [B, total_items]."""
        U, V, b = self._tables()
        return rt.score_all_items("l2", U, V, b, _ids(user_id), device=True)
