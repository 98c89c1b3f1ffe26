JSMPEG_KBENCH_CONFIG=cfg4_2160p python tools/kbench.py 16 24 4 | tail -1 | cut -c1-110
JSMPEG_KBENCH_CONFIG=cfg1_720p python tools/kbench.py 1 360 4 | tail -1 | cut -c1-110
