#!/bin/bash
# build the library; run the given command on a GPU box only if the build succeeded (dev helper)
cd /root/repo
python -m sniffles_amd.build > /tmp/w/build.log 2>&1 || { grep -E "error" -A4 /tmp/w/build.log | head -20; echo "BUILD FAILED"; exit 1; }
/usr/local/graft/bin/gpurun --timeout ${TMO:-900} -- "$@"
