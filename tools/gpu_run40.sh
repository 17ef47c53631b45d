export TMPDIR=/tmp
cd /tmp
rocprofv3 --list-avail 2>/dev/null | grep -E "^\s*(Name|name)|Counter_Name|Metric" | head -5
rocprofv3 -L 2>/dev/null | grep -i -E "MemUnitBusy|MemUnitStalled|L2CacheHit|VALUBusy|TA_BUSY|TCP_|TCC_HIT|TCC_MISS|TCC_REQ|WriteUnitStalled|FetchSize|VALUUtil|TA_TA_BUSY|TCP_TCC_READ|TD_BUSY|SQ_WAIT_INST|SQ_BUSY_CYCLES|SQ_WAVE_CYCLES|SQ_INSTS_VALU\b|GRBM_GUI_ACTIVE|OccupancyPercent|MeanOccupancy" | cut -c1-150 | sort | uniq | head -60
