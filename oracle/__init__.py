"""TEST INFRASTRUCTURE ONLY.

CPU restatement of the CounTR SupervisedMAE hot path (reference: models_mae_cross.py,
models_crossvit.py, util/pos_embed.py, timm 0.4.9 PatchEmbed/Block) used as the parity oracle.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package; the
product path (countr_amd/) never does and fails loudly when the HIP library is missing.

Parity status: PINNED against outputs of the reference itself, run in the build container through
tools/oracle/make_golden.py (fixtures under tests/golden/); the reference ships no tests of its own.
"""
