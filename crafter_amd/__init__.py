"""crafter_amd: MI355X-native batched Crafter environment (the hot path of danijar/crafter).

``Env`` mirrors ``crafter.Env``; ``BatchedEnv`` is its tensor form.  Both need the HIP extension
(crafter_amd/_lib/libcrafter_hip.so, built by ``python -m crafter_amd.build``) and a GPU.
"""
from .batched import BatchedEnv, CrafterDeviceError  # noqa: F401
from .env import Env  # noqa: F401
from .lib import CrafterLibError  # noqa: F401
