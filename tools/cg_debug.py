"""Debug driver for sug_amd.call_graphs: a few steps of the reference-style loop with call graphs on (one model)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
os.environ.setdefault('SUG_CALL_GRAPHS_STRICT', '1')
import test_gpu_call_graphs as T
model = sys.argv[1] if len(sys.argv) > 1 else 'Pointnet'
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4
N = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
a, rng_a, _, _ = T._run(model, B, N, 4, False)
print('eager ok', [x[0] for x in a], flush=True)
b, rng_b, stats, (mgr, net) = T._run(model, B, N, 4, True)
print('graph ok', [x[0] for x in b], stats, flush=True)
print('losses equal', [x[0] for x in a] == [x[0] for x in b], 'state equal', [x[1] for x in a] == [x[1] for x in b], 'rng equal', torch.equal(rng_a, rng_b))
for k, ks in mgr.keys.items():
    print('  key flags', k[0], 'instances', [(id(i) % 10000, None if i.dep is None else id(i.dep) % 10000, i.generation, i.busy) for i in ks.instances], ks.why)
