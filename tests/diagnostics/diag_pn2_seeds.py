"""PointNet++ gradient error vs the fp64 oracle over many seeds, HIP path and fp32 oracle (diagnostic behind
test_pointnet2_gradient_error_statistic_over_8_seeds).  usage: [SUG_SA_FIRST=0] [SUG_POINTMLP_MAX=0] python tests/diagnostics/diag_pn2_seeds.py [NSEEDS] [B]"""
import math, os, statistics, sys
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import test_gpu_model as T

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
rows = []
for i in range(n):
    e_gpu, e_ref, *_ = T._gradient_errors('Pointnet2', 40 + i, 140 + i, B=B, verbose=False)
    rows.append((e_gpu, e_ref))
r = [a / max(b, 1e-12) for a, b in rows]
print('env SA_FIRST=%s POINTMLP_MAX=%s  seeds %d  B %d' % (os.environ.get('SUG_SA_FIRST', '1'), os.environ.get('SUG_POINTMLP_MAX', '1'), n, B))
print('  HIP errors   :', ' '.join('%.1e' % a for a, _ in rows))
print('  fp32 oracle  :', ' '.join('%.1e' % b for _, b in rows))
print('  ratio median %.2f  geometric mean %.2f  max %.1f  min %.2f  #(ratio > 1) %d / %d' % (
    statistics.median(r), math.exp(sum(math.log(x) for x in r) / len(r)), max(r), min(r), sum(x > 1 for x in r), len(r)))
print('  median HIP error %.2e, median fp32-oracle error %.2e' % (statistics.median(a for a, _ in rows), statistics.median(b for _, b in rows)))
