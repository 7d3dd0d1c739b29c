cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
python -m pytest tests/test_hip_groups.py tests/test_hip_parity.py -q -m gpu -p no:cacheprovider --timeout 600 -k "pipelined or forced_handover or poisoned or groups or run_group" 2>&1 | grep -n "^FAILED\|passed\|failed" | head -40
