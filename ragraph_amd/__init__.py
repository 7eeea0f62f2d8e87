"""ragraph_amd -- MI355X-native retrieve-and-propagate hot path of RAGraph (HIP kernels behind a C ABI).

Layout: csrc/ (HIP kernels + C ABI, include/ragraph_hip.h), _native.py (ctypes binding), kernels.py (torch.Tensor
front end), and the host-side mirror of the reference's call surface (ragraph_utils/, layers/, preprompt, RAGraph*).
"""
__version__ = "0.1.0"
