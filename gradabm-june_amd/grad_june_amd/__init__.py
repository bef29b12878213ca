"""grad_june_amd - MI355X-native infection message-passing path behind GradABM-JUNE's Python API.

Exports the reference package's names (grad_june/__init__.py:1-9) so that
``import grad_june_amd as grad_june`` is the switch.  Worlds pickled by the reference load through
:func:`grad_june_amd.graph.load_world`; ``GraphLoader`` / ``AgentDataLoader`` build them from a JUNE
HDF5 file (vectorised; needs ``h5py`` for ``.h5`` input).
"""
from .graph import HeteroData, ToUndirected, load_world, save_world  # noqa: F401
from .infection import IsInfectedSampler  # noqa: F401
from .june_world_loader import AgentDataLoader, GraphLoader  # noqa: F401
from .infection_networks import InfectionNetworks  # noqa: F401
from .model import GradJune  # noqa: F401
from .policies import Policies  # noqa: F401
from .runner import Runner  # noqa: F401
from .symptoms import SymptomsUpdater  # noqa: F401
from .timer import Timer  # noqa: F401
from .transmission import TransmissionSampler, TransmissionUpdater  # noqa: F401

__version__ = "0.1.0"
