#!/bin/bash
export TMPDIR=/tmp
timeout 300 python scripts/dbg_api.py plain 2>/dev/null
timeout 400 python scripts/dbg_api.py with_batch 2>/dev/null
