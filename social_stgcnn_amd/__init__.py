"""social_stgcnn_amd: MI355X-native (gfx950) Social-STGCNN hot path.

Drop-in surface (same names / constructors / state_dict keys as the reference):
    from social_stgcnn_amd.model import social_stgcnn, st_gcn, ConvTemporalGraphical
    from social_stgcnn_amd.utils import seq_to_graph, anorm
    from social_stgcnn_amd.metrics import bivariate_loss

Every compute entry point runs hand-written HIP kernels through the C-ABI library
social_stgcnn_amd/csrc/libstgcnn_hip.so (include/stgcnn_hip.h).  There is no CPU or
eager-PyTorch fallback: a missing library or a non-GPU tensor raises.
"""
__version__ = "0.1.0"
