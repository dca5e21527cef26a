"""HBM traffic per launch of the hot kernels from rocprofv3 PMC counters, written to profiles/pmc_traffic_<family>_<hash>.json
where <hash> = sha256 of the kernel family's sources (bench.kernel_source_hash): bench.py reports `roofline.traffic` only
from the file that matches the sources it runs, so a number of an older kernel version cannot be reported by accident.

Collection as MI355X_MICROARCH.md prescribes: one counter per pass (FETCH_SIZE, WRITE_SIZE do not fit one pass), with
--kernel-trace only; gfx950 correction: FETCH_SIZE tallies the 128-byte requests of wide coalesced reads at 64 bytes -> x2.
Run on the GPU box from the repo root, from a process that has not touched the GPU:  python3 tools/pmc_traffic.py
"""
import collections, csv, glob, json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench

# bench kernel name -> (run_layer_once.py arguments, kernel-name substring in the trace, shape)
CASES = {
    'knn_C64': (['knn', '64'], 'knn_pc_kernel<64', {'B': 64, 'N': 1024, 'k': 20, 'C': 64}),
    'knn_C128': (['knn', '128'], 'knn_pc_kernel<128', {'B': 64, 'N': 1024, 'k': 20, 'C': 128}),
    'knn_C3': (['knn', '3'], 'knn_pc_kernel<4', {'B': 64, 'N': 1024, 'k': 20, 'C': 3}),
    'edgeconv_fused_fwd_C64_Co64': (['edgeconv_fwd', '64', '64'], 'edgeconv_fused_fwd_kernel<64', {'B': 64, 'N': 1024, 'k': 20, 'C': 64, 'Co': 64}),
    'edgeconv_fused_fwd_C64_Co128': (['edgeconv_fwd', '64', '128'], 'edgeconv_fused_fwd_kernel<64', {'B': 64, 'N': 1024, 'k': 20, 'C': 64, 'Co': 128}),
}


def one_pass(counter, args, tag):
    d = '/tmp/pmc_%s_%s' % (tag, counter)
    shutil.rmtree(d, ignore_errors=True)
    env = dict(os.environ, TMPDIR='/tmp')
    cmd = ['rocprofv3', '--pmc', counter, '--kernel-trace', '--output-format', 'csv', '-d', d, '-o', 'p', '--',
           sys.executable, os.path.join(ROOT, 'tools', 'run_layer_once.py')] + args
    subprocess.run(cmd, check=True, cwd='/tmp', env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=300)
    f = glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True)[0]
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r.get('Counter_Name') == counter:
            agg[r['Kernel_Name']].append(float(r['Counter_Value']))
    shutil.rmtree(d, ignore_errors=True)
    return agg


def main():
    out = collections.defaultdict(dict)
    for name, (args, needle, shape) in CASES.items():
        vals = {}
        for ctr in ('FETCH_SIZE', 'WRITE_SIZE'):
            agg = one_pass(ctr, args, name)
            hit = [v for k, v in agg.items() if needle in k]
            if not hit:
                print('no dispatch of %s in the trace of %s' % (needle, args), file=sys.stderr)
                break
            print(name, ctr, 'per dispatch (KB):', [round(v, 1) for v in hit[0]], flush=True)
            vals[ctr] = sorted(hit[0])[len(hit[0]) // 2]          # KB per dispatch: median of the run's dispatches
        else:
            rec = dict(shape)
            rec.update({'fetch_size_kb': vals['FETCH_SIZE'], 'write_size_kb': vals['WRITE_SIZE'],
                        'traffic_bytes': int(round((2 * vals['FETCH_SIZE'] + vals['WRITE_SIZE']) * 1024)),
                        'algorithmic_bytes': bench.kernel_model(name, dict(shape, train=0))['bytes']})
            out[bench.kernel_source_hash(name)][name] = rec
            print(name, rec, flush=True)
    for tag, recs in out.items():
        recs['_comment'] = 'HBM bytes per launch, rocprofv3 --pmc: 2*FETCH_SIZE + WRITE_SIZE (KB -> bytes); tools/pmc_traffic.py; sources hash in the file name'
        for dst in (os.path.join(ROOT, 'gpurun_out'), os.path.join(ROOT, 'profiles')):
            os.makedirs(dst, exist_ok=True)
            json.dump(recs, open(os.path.join(dst, 'pmc_traffic_%s.json' % tag), 'w'), indent=1)


if __name__ == '__main__':
    main()
