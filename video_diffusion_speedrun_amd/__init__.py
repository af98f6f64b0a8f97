"""MI355X-native video-DiT train-step hot path (hand-written HIP for gfx950 behind a C ABI).

Drop-in host API of the reference's model.py / train.py hot path:
    from video_diffusion_speedrun_amd import DiT, apply_fsdp, forward
The package has no CPU / eager-torch compute fallback: importing the compute modules without
the built `libvds_hip.so` raises.
"""
__all__ = ["DiT", "DiTBlock", "apply_fsdp", "forward", "MuAdamW", "generate_latents", "GraphedTrainStep"]


def __getattr__(name):  # lazy: `import video_diffusion_speedrun_amd` itself stays light
    if name in ("DiT", "DiTBlock", "PatchEmbed", "RMSNorm", "ThreeDimRotary", "timestep_embedding"):
        from . import model
        return getattr(model, name)
    if name in ("apply_fsdp", "get_device_mesh"):
        from . import fsdp
        return getattr(fsdp, name)
    if name in ("forward", "train_step"):
        from . import train
        return getattr(train, name)
    if name == "generate_latents":
        from . import sampling
        return sampling.generate_latents
    if name == "GraphedTrainStep":
        from . import graph
        return graph.GraphedTrainStep
    if name == "MuAdamW":
        from . import optim
        return optim.MuAdamW
    raise AttributeError(name)
