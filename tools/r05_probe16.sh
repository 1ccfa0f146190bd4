#!/bin/bash
cd $GRAFT_REPO_ROOT; timeout -k 10 600 python3 tools/probe_deep.py 2>&1 | tail -12
