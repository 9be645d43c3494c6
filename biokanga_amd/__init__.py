"""biokanga_amd - MI355X-native implementation of the `biokanga align` hot path.

The product is the C-ABI shared library `biokanga_amd/lib/libbiokanga_amd.so` (hand-written HIP
kernels for gfx950 + C++ host code, see include/biokanga_amd.h) and the C++ command line front end
`biokanga_amd/bin/biokanga` (`index` / `align`).  This Python package is only plumbing for tests and
bench.py: a ctypes binding of the C ABI.  There is no CPU fallback anywhere - if the library is
missing, or no HIP device is present, calls fail loudly.
"""
from .binding import (Aligner, AlignParams, PEParams, CTX_WINDOW_ARRAY_EAGER, CTX_LEAN_IMAGE, CTX_NO_DEEP_KEYS, CTX_GROW_IMAGE, HIT_DTYPE, LOCI_DTYPE, LOCI_TRIMS_DTYPE, SEG2_DTYPE, SNP_ALN_DTYPE, SNP_SITE_DTYPE, ENTRY_DTYPE, BkError, lib_path, load_library,
                      device_count, build_sa_device, NAR_TAGS, Stream, host_array, seq_counts_allreduce, pack_reads, NBASE_DTYPE, image_policy, POLICY_MIN_READS)

__all__ = ["Aligner", "AlignParams", "PEParams", "HIT_DTYPE", "LOCI_DTYPE", "LOCI_TRIMS_DTYPE", "SEG2_DTYPE", "SNP_ALN_DTYPE", "SNP_SITE_DTYPE", "ENTRY_DTYPE", "BkError", "lib_path", "load_library",
           "device_count", "build_sa_device", "NAR_TAGS", "Stream", "host_array", "seq_counts_allreduce", "pack_reads", "NBASE_DTYPE", "image_policy", "POLICY_MIN_READS"]
