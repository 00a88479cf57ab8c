"""janusx_amd -- MI355X (gfx950) implementation of JanusX's mixed-model hot path.

GRM (ZZ^T) -> eigendecomposition -> null REML (Brent) -> per-SNP LMM Wald scan, as hand-written HIP kernels
behind a C ABI (include/jxgpu.h, janusx_amd/libjxgpu.so).

* ``janusx_amd.janusx``   : drop-in mirror of the reference's native module ``janusx.janusx`` for this path.
* ``janusx_amd.pipeline`` : HBM-resident end-to-end pipeline (what bench.py times).
* ``janusx_amd.stats``    : per-SNP QC/LUT logic on integer counts (bit-exact SNP set).
* ``janusx_amd.bed``      : PLINK payload IO + synthetic panels.
"""
__version__ = "0.1.0"
