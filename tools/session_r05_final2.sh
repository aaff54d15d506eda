#!/bin/bash
# round 5, final session 2: rocprofv3 kernel statistics of the bench command (configs[2], one stream, no secondary legs), PMC FETCH_SIZE / WRITE_SIZE passes of the quarter
# workload (separate passes, counters only), the full k = 55 and k = 127 legs
bash tools/gpu_session.sh r05z2 prof pmck:27:FETCH_SIZE:A=1 pmck:27:WRITE_SIZE:A=1 pmck:55:FETCH_SIZE:A=1 pmck:55:WRITE_SIZE:A=1 k:55 k:127
