#!/bin/bash
# usage: tests/refsuite/run.sh <path under /root/reference/tests> [pytest args...]   (see pyro_alias_plugin.py)
mkdir -p /tmp/refsuite && cd /tmp/refsuite
f=$1; shift
PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=/root/repo/tests/refsuite:/root/repo:/root/reference \
  python -m pytest -p pyro_alias_plugin -p no:cacheprovider --rootdir=/tmp/refsuite -q /root/reference/tests/$f "$@"
