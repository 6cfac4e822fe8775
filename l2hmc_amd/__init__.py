"""l2hmc_amd -- MI355X-native L2HMC leapfrog path behind the reference's
Dynamics / GaugeDynamics operator surface."""
from . import _lib  # noqa: F401
from .lattice import GaugeLattice, u1_plaq_exact, u1_observables  # noqa: F401
from .network import ConvNet3D, GenericNet, MLPNet, network  # noqa: F401
from .gauge_dynamics import GaugeDynamics  # noqa: F401
from .dynamics import Dynamics  # noqa: F401
from .sampler import propose, tf_accept  # noqa: F401
from .distributions import GMM, Gaussian, gen_ring, quadratic_gaussian  # noqa: F401
from .gauge_sampler import GaugeSampler  # noqa: F401
from .gauge_trainer import GaugeTrainer  # noqa: F401
from .dynamics_trainer import DynamicsTrainer  # noqa: F401
from . import stats  # noqa: F401
