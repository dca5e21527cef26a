"""TunableOp pass over the GEMM shapes of one more backbone (the table of tools/tune_gemms.py covers the DGCNN step only):
runs a few eager SUG steps of MODEL with tuning on and writes the chosen solutions to OUT.csv; tools/merge_tunable.py adds
them to sug_amd/tuning/tunableop_gfx950.csv.  usage: python tools/tune_gemms_model.py MODEL BATCH NPOINTS [fp16|caller] OUT.csv   (caller: the unchanged-caller form, four
separate half-batch passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
model, B, N = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
fp16 = len(sys.argv) > 5 and sys.argv[4] == 'fp16'
caller = len(sys.argv) > 5 and sys.argv[4] == 'caller'
out = sys.argv[-1]
os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_TUNING='1', PYTORCH_TUNABLEOP_FILENAME=out,
                  PYTORCH_TUNABLEOP_MAX_TUNING_DURATION_MS=os.environ.get('SUG_TUNE_MS', '15'), PYTORCH_TUNABLEOP_MAX_WARMUP_DURATION_MS='2')
import torch
import torch.cuda.tunable as tn
from bench import synth, BENCH_METHODS
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
tn.set_filename(out, insert_device_ordinal=False)
if fp16:
    from sug_amd.model import Ptran_transformer as PT
    PT.GEMM_DTYPE, PT.PROJ_16BIT = torch.float16, True
torch.manual_seed(666)
tr = SUGStep(Net_MDA(model).cuda().train(), lr=1e-3, weight_decay=5e-5, methods=BENCH_METHODS,
             **({'pair_domains': False, 'share_prefix': False} if caller else {}))
if caller:
    tr.model.g.share_prefix = 'auto'
data = synth(B, N, 666, 'cuda')
for i in range(3):
    tr.step(*data)
    torch.cuda.synchronize()
    print('step', i, 'tuned entries so far', len(tn.get_results()), flush=True)
if hasattr(tn, 'write_file'):
    tn.write_file(out)
print('wrote', out, len(tn.get_results()))
