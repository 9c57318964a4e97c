"""Pin the DLRM oracle against torch-autograd golden fixtures.  CPU only."""
import numpy as np
import pytest

from conftest import golden_files, load_golden, rel_err, OPT_KW
from oracle import numpy_oracle as orc
from oracle.dlrm_oracle import DLRMOracle, interaction_pairs

CFG = dict(m_spa=4, ln_emb=[7, 5, 11], ln_bot=[8, 4], ln_top=[16, 8, 1], dense_dim=13)
KW = {
    "compat": dict(reference_compat=True),
    "compatself": dict(reference_compat=True, arch_interaction_itself=True),
    "intended": dict(reference_compat=False),
    "intendedbce": dict(reference_compat=False, loss_func="bce", loss_threshold=0.05, sigmoid_bot=True),
}


def load_into(o, g, prefix, dtype):
    for f in range(len(o.emb)):
        o.emb[f][:] = g[f"{prefix}emb{f}"].astype(dtype)
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l in range(len(layers)):
            layers[l][0][:] = g[f"{prefix}{nm}{l}W"].astype(dtype)
            layers[l][1][:] = g[f"{prefix}{nm}{l}b"].astype(dtype)


def params(o):
    out = {f"emb{f}": e for f, e in enumerate(o.emb)}
    for nm, layers in (("bot", o.bot), ("top", o.top)):
        for l, (W, b) in enumerate(layers):
            out[f"{nm}{l}W"], out[f"{nm}{l}b"] = W, b
    return out


@pytest.mark.parametrize("fname", golden_files("dlrm"))
@pytest.mark.parametrize("dtype,tol", [(np.float64, 3e-7), (np.float32, 2e-5)])
def test_dlrm_oracle_matches_torch_autograd(fname, dtype, tol):
    _, name, optkind = fname[:-4].split("_")
    g = load_golden(fname)
    o = DLRMOracle(dtype=dtype, seed=0, **CFG, **KW[name])
    load_into(o, g, "in_", dtype)
    opt = {"sgd": orc.SGD, "adagrad": orc.Adagrad, "adam": orc.AdamTFSparse}[optkind](**OPT_KW[optkind])
    losses = [o.step(g["dense"], g["sparse"], g["label"], opt) for _ in range(2)]
    assert rel_err(losses, g["losses"]) < tol
    for k, v in params(o).items():
        assert rel_err(v, g["out_" + k]) < tol * 5, k


def test_reference_interaction_bug_is_reproduced():
    """with the defaults the interaction output is identically zero and the
    embedding tables never move."""
    o = DLRMOracle(dtype=np.float32, seed=1, reference_compat=True, **CFG)
    rng = np.random.default_rng(0)
    dense = rng.uniform(0, 3, (16, 13)).astype(np.float32)
    sparse = np.stack([rng.integers(0, n, 16) for n in CFG["ln_emb"]], 1).astype(np.int32)
    c = o.forward(dense, sparse)
    assert (c["R"][:, CFG["m_spa"]:] == 0).all() and c["R"].shape[1] == CFG["m_spa"] + 6
    _, gr = o.loss_and_grads(dense, sparse, np.zeros(16, np.float32))
    assert (gr["emb"] == 0).all()
    I, J = interaction_pairs(4, True, True)
    assert list(zip(I, J))[:5] == [(0, 0), (0, 1), (0, 2), (0, 3), (1, 1)]
