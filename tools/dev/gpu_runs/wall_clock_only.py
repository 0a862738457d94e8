"""bench.wall_clock() alone (no torch import): the per-genome wall-clock block, incl. the record-table VCF writer."""
import json
import sys

sys.path.insert(0, "/root/repo")
import bench
from sniffles_amd import synth
from sniffles_amd.config import SnifflesConfig

tasks = synth.gen_genome(30.0, seed=1)
out = bench.wall_clock(SnifflesConfig(), tasks, 0)
out = bench.wall_clock(SnifflesConfig(), tasks, 0)      # second run: warm
print(json.dumps(out))
