for cfg in "$@"; do
  echo "== $cfg"; env $cfg timeout 600 python scripts/transcode_profile.py 256 10 seams 2>/dev/null | tail -1
done
