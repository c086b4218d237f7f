# A/B of the 1024-file transcode under environment switches (round 5): bash scripts/experiments/transcode_ab.sh "A=0" "DMI_NO_TAPER=1" ...
for cfg in "$@"; do
  echo "== $cfg"; env $cfg timeout 600 python scripts/transcode_profile.py 1024 8 2>/dev/null | tail -1
done
