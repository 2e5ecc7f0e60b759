"""Host-side helpers that pin the vendor libraries' kernel choices (tuning.py): no GPU needed."""
import os

import torch


def test_gemm_tuning_is_a_no_op_without_a_gpu(gfla, tmp_path):
    if torch.cuda.is_available():
        return
    assert gfla.enable_gemm_tuning(str(tmp_path / "t.csv")) is False
    assert not (tmp_path / "t.csv").exists()
    assert gfla.gemm_tuning_results() == []


def test_shipped_results_files_are_well_formed(gfla):
    from global_flow_local_attention_amd import tuning
    lines = open(tuning.SHIPPED_RESULTS).read().strip().splitlines()
    validators = [l for l in lines if l.startswith("Validator,")]
    entries = [l for l in lines if l.startswith("GemmTunableOp_float_")]
    assert len(validators) >= 4 and any("gfx950" in v for v in validators)
    assert len(entries) == 6 and all(len(e.split(",")) == 4 for e in entries)   # 3 GEMMs x 2 attention layers
    names = sorted(os.listdir(tuning.SHIPPED_MIOPEN_DB))
    assert any(n.endswith(".ufdb.txt") for n in names) and any(n.endswith(".udb.txt") for n in names)
    assert all(n.startswith("gfx950") for n in names)


def test_seed_conv_db_copies_once_and_respects_the_user(gfla, tmp_path, monkeypatch):
    monkeypatch.delenv("MIOPEN_USER_DB_PATH", raising=False)
    d = str(tmp_path / "db")
    assert gfla.seed_conv_db(d) == d and os.environ["MIOPEN_USER_DB_PATH"] == d
    copied = sorted(os.listdir(d))
    assert copied and all(n.startswith("gfx950") for n in copied)
    # an existing (possibly grown) database file is never overwritten
    first = os.path.join(d, copied[0])
    with open(first, "a") as f:
        f.write("extra\n")
    monkeypatch.delenv("MIOPEN_USER_DB_PATH")
    gfla.seed_conv_db(d)
    assert open(first).read().endswith("extra\n")
    # a path the user chose wins
    monkeypatch.setenv("MIOPEN_USER_DB_PATH", "/somewhere/else")
    assert gfla.seed_conv_db(str(tmp_path / "other")) == "/somewhere/else"
    assert not (tmp_path / "other").exists()
