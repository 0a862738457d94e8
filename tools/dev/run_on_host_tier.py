"""Run a tools/dev sweep on the host tier (the unchanged HIP sources on the fibre shim) instead of the GPU."""
import runpy, sys
sys.path[:0] = ["/root/repo", "/root/repo/tests", "/root/repo/oracle"]
import emu.emu as E
E.lib()
script = sys.argv[1]; sys.argv = sys.argv[1:]
runpy.run_path(script, run_name="__main__")
