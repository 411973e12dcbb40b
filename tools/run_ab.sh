timeout 600 python -m pytest tests -m gpu -x -q -k "launch_split or edge_embed" 2>&1 | grep -v "^  File\|^Extension\|^$" | tail -5
