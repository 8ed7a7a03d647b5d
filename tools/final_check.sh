timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -2
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
timeout 600 python bench.py 2>/dev/null | tail -1 | cut -c1-700
