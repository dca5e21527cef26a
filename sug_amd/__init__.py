"""sug_amd -- MI355X-native implementation of SUG's point-cloud encoder + MMD alignment hot path.

Drop-in surface (same names / arguments as the reference):
    sug_amd.model.Model.Net_MDA(model_name).forward(x, ..., semantic_adaption=...)
    sug_amd.model.mmd.mmd_cal(label_s, feat_s, label_t, feat_t, args, data_s, data_t)
backed by hand-written HIP kernels in libsug_amd.so (C ABI: include/sug_amd.h).
"""
__version__ = '0.1.0'
