export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -k "physics_loss_vs_reference and lo2d" 2>&1 | grep -E "^E  |assert|passed|failed" | head -12)
