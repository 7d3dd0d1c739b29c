cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
python -m pytest tests -q -m gpu -p no:cacheprovider --timeout 600 -k "${1:-pipelined or forced_handover or poisoned or groups or run_group}" 2>&1 | grep -n "^FAILED\|passed\|failed\|^E  " | head -40
