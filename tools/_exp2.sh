#!/bin/bash
cd $GRAFT_REPO_ROOT
python tools/host_enqueue.py 4 100 1024 2>&1 | tail -3
python tools/host_enqueue.py 1 100 1024 2>&1 | tail -3
python tools/host_threads.py 1 4 100 2>&1 | tail -3
python tools/host_threads.py 2 2 100 2>&1 | tail -3
python tools/host_threads.py 4 1 100 2>&1 | tail -3
python tools/host_threads.py 4 2 100 2>&1 | tail -3
python tools/host_threads.py 8 1 100 2>&1 | tail -3
