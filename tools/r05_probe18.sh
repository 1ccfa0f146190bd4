#!/bin/bash
cd $GRAFT_REPO_ROOT; timeout -k 10 600 python3 tools/probe_capture_after_eager.py 2>&1 | tail -14
