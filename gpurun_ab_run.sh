python -m pytest tests -m gpu -q -x 2>&1 | tail -2
bash profiles/ab_env.sh "--workload refdefault --steps 12 --warmup 2" "-"
