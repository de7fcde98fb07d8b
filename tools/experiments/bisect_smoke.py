import sys, os
sys.path.insert(0, os.getcwd())
which = sys.argv[1]
import __graft_entry__ as g
if which == 'cdll':
    from torchsr_amd import _lib
    import ctypes
    h = ctypes.CDLL(_lib.LIB_PATH)
elif which == 'imports':
    import torchsr_amd, torchsr_amd.srgan.trainer, oracle.srgan
elif which == 'cdll_after_init':
    import torch
    torch.cuda.is_available()
    from torchsr_amd import _lib
    import ctypes
    h = ctypes.CDLL(_lib.LIB_PATH)
elif which == 'build_noload':
    from torchsr_amd import _lib
    _lib.build(force=False, verbose=True)
try:
    g.smoke()
except Exception as e:
    print(which, 'FAILED', str(e)[-120:])
