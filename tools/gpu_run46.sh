export TMPDIR=/tmp
(timeout 1200 python -m pytest tests/test_stage1.py -m gpu -q --timeout=600 2>&1 | grep -E "^FAILED|passed|failed|Error|assert" | head -8)
