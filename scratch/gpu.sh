#!/bin/bash
# helper: rebuild the HIP library locally, then run a command on the GPU box
python maest_amd/build.py 2>&1 | grep -E "error|warning: |built" 
/usr/local/graft/bin/gpurun --timeout ${TMO:-900} -- "$@"
