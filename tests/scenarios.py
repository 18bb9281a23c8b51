"""The workload builders live in the package (theia_amd/workloads.py); the tests use them under this name."""
from theia_amd.workloads import *  # noqa: F401,F403
from theia_amd.workloads import Scenario, run_batches  # noqa: F401
