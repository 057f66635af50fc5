#!/usr/bin/env python3
"""Is the communicator of parallel.capture_group() connected when the group comes back (before any collective)?  FTHMC_FORCE_PG=1 python3 tools/pg_group_check.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
from fthmc_amd import parallel as P
P.init()
g = P.capture_group(torch.device("cuda", 0))
be = g._get_backend(torch.device("cuda", 0))
print("supports_splitting", getattr(be, "supports_splitting", None), "comm initialised", be._is_initialized() if hasattr(be, "_is_initialized") else None, flush=True)
dist.destroy_process_group()
