"""crafter_amd: MI355X-native batched Crafter environment (the hot path of danijar/crafter).

``Env`` mirrors ``crafter.Env``; ``BatchedEnv`` is its tensor form.  Both need the HIP extension
(crafter_amd/_lib/libcrafter_hip.so, built by ``python -m crafter_amd.build``) and a GPU.
"""
from .batched import BatchedEnv, CrafterDeviceError  # noqa: F401
from .env import Env  # noqa: F401
from .lib import CrafterLibError  # noqa: F401
from .recorder import BatchedEpisodeRecorder, BatchedStatsRecorder  # noqa: F401
from .vec import VecEnvView  # noqa: F401

try:  # gym is optional, exactly like the reference (crafter/__init__.py:4-17)
  import gym
  gym.register(id='CrafterAmdReward-v1', entry_point='crafter_amd:Env', max_episode_steps=10000,
               kwargs={'reward': True})
  gym.register(id='CrafterAmdNoReward-v1', entry_point='crafter_amd:Env', max_episode_steps=10000,
               kwargs={'reward': False})
except ImportError:
  pass
