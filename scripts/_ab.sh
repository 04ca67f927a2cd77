for a in 0 1 2 4 3 7 0; do echo "abl $a"; TUP_TAIL_ABLATE=$a timeout -k 10 200 python scripts/stage_times.py 2>&1 | grep -E "tail"; done
