"""Make the real AudibleLight package use the MI355X path.

``Scene.generate`` resolves ``audiblelight.synthesize`` functions by lazy import at call time
(reference core.py:1828-1831,1851,1866), so replacing the module attributes is a true drop-in.
"""
from __future__ import annotations

REPLACED = ("apply_snr", "db_to_multiplier", "time_invariant_convolution", "time_variant_convolution",
            "normalize_irs", "compute_dry_audio", "render_event_audio", "render_audio_for_all_scene_events",
            "generate_scene_audio_from_events", "validate_scene", "generate_interpolation_matrix", "stft",
            "perform_time_variant_convolution", "istft_overlap_synthesis")


def install(module=None):
    """Patch ``audiblelight.synthesize`` (or ``module``) in place; returns the originals for ``uninstall``."""
    from . import synthesize as ours

    if module is None:
        import audiblelight.synthesize as module  # the reference package must be importable
    saved = {}
    for name in REPLACED:
        saved[name] = getattr(module, name, None)
        setattr(module, name, getattr(ours, name))
    module._audiblelight_amd_saved = saved
    return saved


def uninstall(module=None) -> None:
    if module is None:
        import audiblelight.synthesize as module
    for name, fn in getattr(module, "_audiblelight_amd_saved", {}).items():
        if fn is not None:
            setattr(module, name, fn)
