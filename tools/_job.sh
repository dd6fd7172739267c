cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q --timeout 900 2>&1 | grep -E "passed|failed|FAILED|error" | tail -8
