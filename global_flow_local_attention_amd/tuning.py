"""Library-GEMM selection for the FC layers of ExtractorAttn.

The first FC layer runs as plain fp32 GEMMs through torch (`extractor_attn._source_half_fc`): skinny ones,
M = 128 hidden channels against K = C*k^2 and N = B*H*W.  The default heuristics of hipBLASLt pick
75-106 TF/s solutions for them on MI355X; PyTorch's TunableOp, which times every rocBLAS / hipBLASLt
solution once per new shape and remembers the winner, finds 105-126 TF/s ones
(profiles/r1_tunableop_fc_gemms.txt).  This module only switches that mechanism on; the GEMMs stay library
GEMMs.  Tuning costs a few seconds per new GEMM shape the first time it is seen (in a warm-up step) and is
cached in `filename` (validated by torch against the torch / ROCm / hipBLASLt versions that produced it).
"""
import os

import torch


SHIPPED_RESULTS = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "tunableop_gfx950_fc_gemms.csv")


def enable_gemm_tuning(filename=None, max_duration_ms=200, max_iterations=20, tune=True, seed=SHIPPED_RESULTS):
    """Turn TunableOp on for this process.  filename: results cache (read if present, written at exit).
    seed: a results file copied to `filename` when that does not exist yet -- by default the solutions tuned
    on MI355X for the FC GEMMs of the PoseGenerator shapes (tuned/tunableop_gfx950_fc_gemms.csv); torch ignores
    it, and tunes from scratch, when its validator lines (torch / ROCm / hipBLASLt / rocBLAS versions, GPU
    architecture) do not match the running stack.  tune=False only replays a cache.  Returns False when
    torch has no TunableOp (nothing changes then)."""
    tunable = getattr(torch.cuda, "tunable", None)
    if tunable is None or not torch.cuda.is_available():
        return False
    try:
        tunable.enable(True)
        tunable.tuning_enable(bool(tune))
        tunable.set_max_tuning_duration(int(max_duration_ms))
        tunable.set_max_tuning_iterations(int(max_iterations))
        if filename:
            os.makedirs(os.path.dirname(os.path.abspath(filename)), exist_ok=True)
            if seed and os.path.exists(seed) and not os.path.exists(filename):
                import shutil
                shutil.copyfile(seed, filename)
            tunable.set_filename(filename)
    except Exception as e:  # an optional speed-up must never take the caller down
        import warnings
        warnings.warn("TunableOp could not be enabled (%s); GEMMs stay on the default heuristics" % e)
        try:
            tunable.enable(False)
        except Exception:
            pass
        return False
    return True


SHIPPED_MIOPEN_DB = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuned", "miopen")


def seed_conv_db(directory, seed=SHIPPED_MIOPEN_DB):
    """Point MIOpen's user databases (find results + tuned kernel parameters) at `directory`, pre-filled
    with the entries recorded on MI355X for the target-half convolutions of the PoseGenerator shapes
    (tuned/miopen/*.ufdb.txt, *.udb.txt).  MIOpen times each candidate once when torch asks it to `find` an
    algorithm and does not always settle on the same implicit-GEMM configuration (the L2 backward-data
    convolution came out at 658, 674, 689, 739 and 767 us in five runs); with the database present it reuses
    the recorded choice and skips the search.  The file names carry the GPU and the MIOpen version, so a
    different stack simply ignores them.  Must be called before the first convolution of the process; does
    nothing when the user already set MIOPEN_USER_DB_PATH.  Returns the directory in use."""
    if os.environ.get("MIOPEN_USER_DB_PATH"):
        return os.environ["MIOPEN_USER_DB_PATH"]
    try:
        os.makedirs(directory, exist_ok=True)
        if seed and os.path.isdir(seed):
            import shutil
            for name in os.listdir(seed):
                dst = os.path.join(directory, name)
                if not os.path.exists(dst):
                    shutil.copyfile(os.path.join(seed, name), dst)
        os.environ["MIOPEN_USER_DB_PATH"] = directory
        return directory
    except OSError:
        return None


def gemm_tuning_results():
    """The (operator, shape, chosen solution, time) rows TunableOp holds in this process; [] when it is off."""
    tunable = getattr(torch.cuda, "tunable", None)
    if tunable is None or not torch.cuda.is_available():
        return []
    try:
        return list(tunable.get_results()) if tunable.is_enabled() else []
    except Exception:
        return []
