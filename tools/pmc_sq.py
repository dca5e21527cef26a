"""SQ counters of the hot kernels, one counter per rocprofv3 pass (with --kernel-trace only), per-kernel averages over the
dispatches of `python3 tools/run_layer_once.py ARGS`.  usage (GPU box, repo root, from a process that has not touched the GPU):
    python3 tools/pmc_sq.py OUT.txt KERNEL_SUBSTRING run_layer_once-args...      e.g.  ... knn_pc_kernel knn 64"""
import collections, csv, glob, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COUNTERS = ['SQ_WAVES', 'SQ_BUSY_CYCLES', 'SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_WAIT_INST_LDS', 'SQ_ACTIVE_INST_ANY',
            'SQ_ACTIVE_INST_VALU', 'SQ_ACTIVE_INST_LDS', 'SQ_INSTS_VALU', 'SQ_INSTS_MFMA', 'SQ_INSTS_SALU', 'SQ_INSTS_LDS',
            'SQ_VALU_MFMA_BUSY_CYCLES', 'SQ_VALU_MFMA_COEXEC_CYCLES', 'SQ_LDS_BANK_CONFLICT', 'SQ_LDS_IDX_ACTIVE', 'GRBM_GUI_ACTIVE']
out, needle, args = sys.argv[1], sys.argv[2], sys.argv[3:]
res = {}
for ctr in COUNTERS:
    d = '/tmp/pmcsq_%s' % ctr
    shutil.rmtree(d, ignore_errors=True)
    cmd = ['rocprofv3', '--pmc', ctr, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
           sys.executable, os.path.join(ROOT, 'tools', 'run_layer_once.py')] + args
    r = subprocess.run(cmd, cwd='/tmp', env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, timeout=300)
    files = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)
    if r.returncode != 0 or not files:
        res[ctr] = 'failed (%s)' % r.stderr.decode()[-200:].strip().replace('\n', ' ')
        continue
    vals = [float(x['Counter_Value']) for x in csv.DictReader(open(files[0])) if x.get('Counter_Name') == ctr and needle in x['Kernel_Name']]
    res[ctr] = (sum(vals) / len(vals), len(vals)) if vals else 'no dispatch of %s' % needle
    shutil.rmtree(d, ignore_errors=True)
    print(ctr, res[ctr], flush=True)
with open(out, 'w') as f:
    f.write('kernel %s, workload `run_layer_once.py %s`: average per dispatch (one counter per pass, --kernel-trace only)\n' % (needle, ' '.join(args)))
    for c in COUNTERS:
        v = res[c]
        f.write('%-28s %s\n' % (c, ('%16.1f  (%d dispatches)' % v) if isinstance(v, tuple) else v))
