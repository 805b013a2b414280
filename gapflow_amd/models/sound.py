"""`GaPFlow.models.sound.eos_sound_velocity` (sound.py:35-81) as a device operator: c = sqrt(dp/drho)."""
from .pressure import _eos_call


def eos_sound_velocity(density, prop):
    """Local speed of sound of a density field for the equation of state named in prop['EOS']."""
    return _eos_call(density, prop, True)
