for n in 4 8; do
  python tools/dev/search_mode_cost.py $n 34 4 20
  python tools/dev/search_mode_cost.py $n 34 64 20
  python tools/dev/search_mode_cost.py $n 40 64 20
  python tools/dev/search_mode_cost.py $n 60 64 20
  python tools/dev/search_mode_cost.py $n 10 64 20
  python tools/dev/search_mode_cost.py $n 0 64 20
  python tools/dev/search_mode_cost.py $n 1 64 20
  python tools/dev/search_mode_cost.py $n 50 64 20
done
