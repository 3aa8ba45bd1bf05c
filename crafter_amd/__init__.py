"""crafter_amd: MI355X-native batched Crafter environment (the hot path of danijar/crafter).

``Env`` mirrors ``crafter.Env``; ``BatchedEnv`` is its tensor form.  Both need the HIP extension
(crafter_amd/_lib/libcrafter_hip.so, built by ``python -m crafter_amd.build``) and a GPU.
"""
from .batched import BatchedEnv, CrafterDeviceError  # noqa: F401
from .env import Env  # noqa: F401
from .lib import CrafterLibError  # noqa: F401
from .recorder import BatchedEpisodeRecorder, BatchedStatsRecorder, EnvStatsRecorder  # noqa: F401
from .vec import VecEnvView  # noqa: F401

try:  # gym is optional, exactly like the reference (crafter/__init__.py:4-17)
  import gym
  gym.register(id='CrafterAmdReward-v1', entry_point='crafter_amd:Env', max_episode_steps=10000,
               kwargs={'reward': True})
  gym.register(id='CrafterAmdNoReward-v1', entry_point='crafter_amd:Env', max_episode_steps=10000,
               kwargs={'reward': False})
except ImportError:
  pass


def register_reference_ids(force=False):
  """Opt-in drop-in: claim the reference's gym ids ``CrafterReward-v1`` / ``CrafterNoReward-v1``
  (crafter/__init__.py:4-17) for ``crafter_amd.Env``, so that ``gym.make('CrafterReward-v1')`` in existing agent
  code builds the MI355X env.  Not done at import: a process that also imports the reference ``crafter`` package
  would otherwise see two registrations of the same id.  ``force`` replaces an existing registration (gym versions
  differ in whether they raise, warn or overwrite on a duplicate id).  Returns the ids registered; raises
  ImportError without gym, like ``gym.make`` itself would."""
  import gym
  done = []
  for name, reward in (('CrafterReward-v1', True), ('CrafterNoReward-v1', False)):
    registry = getattr(gym.envs.registration, 'registry', None)
    specs = getattr(registry, 'env_specs', registry)   # old gym: EnvRegistry.env_specs, new gym: a dict
    if specs is not None and name in specs:
      if not force:
        continue
      del specs[name]
    gym.register(id=name, entry_point='crafter_amd:Env', max_episode_steps=10000, kwargs={'reward': reward})
    done.append(name)
  return done
