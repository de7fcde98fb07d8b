"""Developer switches (same-box A/B runs): read from the environment ONCE, at import -- never per forward call.
The release behaviour is every switch off."""
import os


def _on(name: str) -> bool:
    return os.environ.get(name, '') not in ('', '0')


NO_ACT_FOLD = _on('SRX_NO_ACT_FOLD')            # keep the first LeakyReLU backward of the discriminators a pass of its own
NO_BN_DGRAD_FUSE = _on('SRX_NO_BN_DGRAD_FUSE')  # residual tower: separate BatchNorm-backward reduce launches
NO_RDB_FUSED = _on('SRX_NO_RDB_FUSED')          # ESRGAN dense blocks: five conv launches instead of srx_rdb_fwd / srx_rdb_bwd
NO_C64 = _on('SRX_NO_C64')                      # bf16 inference: fp32-stored activations (round 3's path) instead of the bf16-native chain
NO_T9 = _on('SRX_NO_T9')                        # bf16 inference: the 9x9 output conv on the 4x4x4 MFMA kernel instead of thin9.hip
NO_WINO = _on('SRX_NO_WINO')                    # VGG19's wide 3x3 layers on the direct gather-GEMM instead of Winograd F(2x2, 3x3)
NO_BF16S = _on('SRX_NO_BF16S')                  # frozen conv stacks under autocast: fp32-stored activations (round 4's path) instead of bf16 storage
NO_OVERLAP = _on('SRX_NO_OVERLAP')              # single-graph GAN step: the perceptual-loss branch on the main stream (no second graph branch)
FWD_OVERLAP_ONLY = _on('SRX_FWD_OVERLAP_ONLY')  # ... only the perceptual loss's FORWARD on the side stream (its backward where autograd puts it)
