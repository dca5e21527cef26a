"""Regenerate sug_amd/tuning/tunableop_gfx950.csv: run a few C2 training steps with TunableOp tuning
enabled (needs an MI355X).  usage: python tools/tune_gemms.py [out.csv]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = sys.argv[1] if len(sys.argv) > 1 else 'gpurun_out/tunableop_gfx950.csv'
os.environ.update(PYTORCH_TUNABLEOP_ENABLED='1', PYTORCH_TUNABLEOP_TUNING='1', PYTORCH_TUNABLEOP_FILENAME=out)
import torch
import torch.cuda.tunable as tn
from bench import synth
from sug_amd.model.Model import Net_MDA
from sug_amd.train_step import SUGStep
tn.set_filename(out, insert_device_ordinal=False)
dev = torch.device('cuda')
torch.manual_seed(666)
tr = SUGStep(Net_MDA('DGCNN').to(dev).train(), lr=1e-3, weight_decay=5e-5)
data = synth(32, 1024, 666, dev)
for _ in range(4):
    tr.step(*data)
torch.cuda.synchronize()
tn.write_file(out) if hasattr(tn, 'write_file') else None
print('tuned %d GEMM shapes -> %s' % (len(tn.get_results()), out))
