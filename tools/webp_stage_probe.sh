for d in 0 1 3 7; do CSH_WEBP_DEBUG=$d python tools/mixed_probe.py 96 webp 2>&1 | grep -E "^webp|cli\]" | head -2; done
