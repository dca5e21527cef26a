"""SHA-256 of the neighbour lists of sug_knn on seeded inputs (run on a GPU box; the committed
knn_pc_hashes.json was produced by the exact-list consumer of round 2, commit 'EdgeConv forward: both domain
groups in one launch', whose lists the other kNN tests pin against the oracle).  The packed-key consumer that
replaced it must reproduce them bit for bit on every path (fast / exact re-rank / full rescan).
usage: python tests/golden/make_knn_hashes.py out.json"""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch


def cases():
    """name -> x [B,N,C] float32 (CPU generator: identical on every machine)"""
    out = {}
    for C, N, B, seed in ((64, 1024, 4, 1), (128, 1024, 3, 2), (3, 1024, 4, 3), (64, 2048, 2, 4), (128, 600, 2, 5),
                          (3, 2048, 2, 6), (64, 96, 3, 7)):
        g = torch.Generator().manual_seed(seed)
        out['randn_C%d_N%d' % (C, N)] = torch.randn(B, N, C, generator=g)
        # clustered features with a common offset (small relative gaps between neighbour distances)
        out['clustered_C%d_N%d' % (C, N)] = torch.randn(B, N, C, generator=g) * 0.05 + torch.randn(B, 1, C, generator=g) * 3
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 1024, 64, generator=g)
    x[:, 512:] = x[:, :512]                                   # every point has an exact duplicate
    out['duplicates_C64'] = x
    x = torch.randn(2, 1024, 3, generator=g)
    x[:, 100:400] = x[:, 7:8]                                  # 300 copies of one point (padded cloud)
    out['padded_C3'] = x
    gx = torch.stack(torch.meshgrid(torch.arange(16.), torch.arange(8.), torch.arange(8.), indexing='ij'), -1).reshape(1, 1024, 3)
    out['lattice_C3'] = gx * 0.125                             # exact ties everywhere
    gf = torch.zeros(1, 1024, 64)
    gf[0, :, :3] = gx[0]
    out['lattice_C64'] = gf
    x = torch.randn(1, 1024, 128, generator=g)
    x[0, ::2] = x[0, 0]                                        # half of the cloud is one point
    out['half_identical_C128'] = x
    return out


def digest(idx):
    return hashlib.sha256(idx.cpu().to(torch.int32).contiguous().numpy().tobytes()).hexdigest()


if __name__ == '__main__':
    from sug_amd import ops
    res = {}
    for name, x in cases().items():
        for k in (20, 16):
            res['%s_k%d' % (name, k)] = digest(ops.knn(x.cuda(), k))
    json.dump(res, open(sys.argv[1], 'w'), indent=0, sort_keys=True)
    print(len(res), 'hashes')
