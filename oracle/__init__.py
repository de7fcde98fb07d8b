"""ORACLE -- test infrastructure only.

CPU restatement of the roclark/torchsr hot path (plain ``torch.nn.functional`` on
NCHW fp32), pinned against the reference's own modules by ``oracle/gen_golden.py``.
The product (``torchsr_amd``) never imports this package.
"""
