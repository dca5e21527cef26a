"""CPU: the C-ABI library loads and exports exactly what include/sug_amd.h declares."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    src = open(os.path.join(ROOT, 'include', 'sug_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    names = re.findall(r'\b(?:int|int64_t|const char\*)\s+(sug_[a-z0-9_]+)\s*\(', src)
    return sorted(set(names))


def _count_args(name):
    src = open(os.path.join(ROOT, 'include', 'sug_amd.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    m = re.search(r'\b' + name + r'\s*\((.*?)\)\s*;', src, flags=re.S)
    args = m.group(1).strip()
    return 0 if args == 'void' else len(args.split(','))


def test_library_built_and_exports_header_symbols():
    from sug_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), 'build with make -C sug_amd/csrc'
    L = ctypes.CDLL(_lib.LIB_PATH)
    names = _declared()
    assert len(names) >= 18
    for n in names:
        assert hasattr(L, n), 'missing export %s' % n


def test_ctypes_table_matches_header():
    from sug_amd import _lib
    names = [n for n in _declared() if n not in ('sug_last_error', 'sug_abi_version', 'sug_linear_dw_workspace', 'sug_adam_chunk', 'sug_adam_chain_chunk',
                                               'sug_pointmlp_max_bwd_workspace', 'sug_ptran_colsum_workspace', 'sug_colsum_workspace',
                                               'sug_chamfer_workspace', 'sug_scatter_rows_workspace')]
    assert sorted(_lib.SIGNATURES) == names
    for n in names:
        assert len(_lib.SIGNATURES[n]) == _count_args(n), n


def test_version_and_error_string():
    from sug_amd import _lib
    L = _lib.lib()
    assert L.sug_abi_version() == 7
    # argument validation happens on the host before any launch: safe without a GPU
    rc = L.sug_knn(None, 3, 1, 8, 3, 4, None, None)
    assert rc == -1 and b'null' in L.sug_last_error()
    rc = L.sug_knn(ctypes.c_void_p(16), 3, 1, 8, 3, 40, ctypes.c_void_p(16), None)
    assert rc == -1 and b'k' in L.sug_last_error()


def test_ops_refuse_cpu_tensors():
    import torch
    from sug_amd import ops
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.knn(torch.zeros(1, 8, 3), 4)
    with pytest.raises(RuntimeError, match='HIP device'):
        ops.mix_rbf_mmd2_rows(torch.zeros(4, 8), 2)


def test_tuning_tables_are_well_formed():
    """sug_amd/tuning: the TunableOp table has validator lines + GEMM entries, and every weight-gradient
    shape routed to the library has a tuned TN entry with K = rows."""
    import csv
    import json
    from sug_amd import tuning
    rows = list(csv.reader(open(tuning.TABLE)))
    assert any(r[0] == 'Validator' and r[1] == 'GCN_ARCH_NAME' and r[2].startswith('gfx950') for r in rows)
    gemms = [r for r in rows if r[0].startswith('Gemm')]
    assert len(gemms) >= 20
    choice = json.load(open(tuning.DW_CHOICE))['library']
    keys = {r[1] for r in gemms}
    for R, M, N in choice:
        assert any(k.startswith('nt_%d_%d_%d_' % (N, M, R)) or k.startswith('tn_%d_%d_%d_' % (N, M, R)) or
                   ('_%d_' % R) in k for k in keys), (R, M, N)
