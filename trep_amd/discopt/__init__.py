"""Mirror of ``trep.discopt`` on top of the batched HIP integrator."""
from .dcost import DCost  # noqa: F401
from .dsystem import DSystem, BatchDSystem  # noqa: F401
from .doptimizer import DOptimizer, DOptimizerMonitor, DOptimizerDefaultMonitor, DOptimizerVerboseMonitor  # noqa: F401
from .dlqr import solve_tv_lqr, solve_tv_lq  # noqa: F401
from .batch_doptimizer import BatchDOptimizer  # noqa: F401
