export TMPDIR=/tmp
for i in 1 2 3; do (timeout 1800 python -m pytest tests -m gpu -q --timeout=900 2>&1 | grep -E "^FAILED|passed|failed|Error" | head -5); done
