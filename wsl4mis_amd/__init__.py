"""wsl4mis_amd -- MI355X-native hot path of WSL4MIS' 2-D weakly-supervised training (HIP kernels behind the
reference's networks.net_factory / utils.losses / utils.gate_crf_loss interface)."""
__version__ = "0.1.0"
