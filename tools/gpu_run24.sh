export TMPDIR=/tmp
(timeout 1800 python -m pytest tests -m gpu -q --timeout=900 -k "stage3 or advective" 2>&1 | tail -15)
