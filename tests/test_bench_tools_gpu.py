"""bench.py's own machinery on a small batch: the north star's op-level roofline leg and the oracle check (max-abs on two
samples) -- so that a typo there does not burn the driver's one bench run."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)


def _hotpath(B=2):
    import bench
    import global_flow_local_attention_amd as gfla
    hp = bench.HotPath(B, DEV, seed=5, fc_impl="mfma", fc_mode=4)
    return bench, hp, gfla.Resample2d(4, 1, 2)


def test_north_star_leg_and_oracle_check_run():
    bench, hp, resample = _hotpath()
    r = bench.op_roofline(DEV, B=2, iters=2)
    for name, d in r["layers"].items():
        for key in ("block_extractor_fwd", "local_attn_fwd", "pair"):
            assert d[key]["us"] > 0 and 0 < d[key]["frac"] < 1.5
    hp.step(resample, allreduce=False)
    chk = bench.oracle_check(hp, resample)
    assert chk["samples"] == [0, 1] and max(chk["max_rel"].values()) <= 1e-4
    assert max(v for k, v in chk["max_abs"].items() if k.endswith(" out") or k.endswith(" warp")) <= 1e-4
