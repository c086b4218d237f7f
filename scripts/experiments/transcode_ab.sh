# A/B of the 1024-file transcode under environment switches (round 5): bash scripts/experiments/transcode_ab.sh
for cfg in "A=0" "DMI_PREPARE_WORKERS=1" "DMI_NO_REGISTER=1" "DMI_NO_INGEST=1" "DMI_BUILD_WORKERS=1" "A=1" "DMI_TRANSCODE_PYTHON=1"; do
  echo "== $cfg"; env $cfg timeout 600 python scripts/transcode_time.py 1024 2>/dev/null | grep -E "\"value\"|ms_per_batch|build \(|prepare \(|encode \(|assemble|register|parse"
done
