"""Tuned library-GEMM selection for the encoders' forward / input-gradient GEMMs.

The dense GEMMs of the path are plain rocBLAS / hipBLASLt calls made by torch.  PyTorch's
TunableOp can pick, per GEMM shape, the fastest solution of either library; `tunableop_gfx950.csv`
holds the choices for the shapes of the C2 workload (DGCNN, 32 clouds per domain, N=1024), recorded
on an MI355X with `tools/tune_gemms.py`, and -- round 4 -- of the other benched workloads (PointNet 8 per domain,
PointNet++ 64 x 2048, Point Transformer 16 x 2048 with fp16 linears, and the unchanged-caller form of the DGCNN step),
recorded with `tools/tune_gemms_model.py` and merged by `tools/merge_tunable.py` (config 5: 20.4 -> 20.0 ms, config 3:
17.5 -> 17.2 ms on the same box; no memset nodes: tools/find_memsets.py).  `enable_tuned_gemms()` switches TunableOp on in look-up
mode (no tuning at run time; unknown shapes fall back to the default heuristic), and routes the
weight gradients of the shapes listed in `dw_choice_gfx950.json` -- where the tuned library GEMM
measured faster than sug_linear_dw -- to the library (those are not bit-reproducible run to run
if the chosen solution accumulates with atomics; the parity tests never enable this).  The validator
lines of the file pin the ROCm / library versions; on a different stack the table is ignored."""
import os
import tempfile

TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'tunableop_gfx950.csv')
DW_CHOICE = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dw_choice_gfx950.json')


def enable_tuned_gemms(table=None):
    """Returns True if the table was loaded (SUG_TUNABLE_TABLE overrides the shipped file: A/B runs)."""
    table = table or os.environ.get('SUG_TUNABLE_TABLE') or TABLE
    import torch.cuda.tunable as tn
    if not os.path.exists(table):
        return False
    tn.enable(True)
    tn.tuning_enable(False)
    # results written at exit go to a scratch file, never into the source tree
    tn.set_filename(os.path.join(tempfile.gettempdir(), 'sug_amd_tunableop_%d.csv' % os.getpid()))
    ok = bool(tn.read_file(table))
    if ok and os.path.exists(DW_CHOICE):
        # weight-gradient shapes where the tuned library GEMM measured faster than sug_linear_dw
        import json
        from .. import ops
        ops.DW_LIBRARY_SHAPES = {tuple(v) for v in json.load(open(DW_CHOICE))['library']}
    return ok
