#!/bin/bash
# dev: collect the observed parity errors of the whole GPU suite (class ceilings apply while collecting) and rewrite the table
rm -f gpurun_out/err_log.txt
EA_TEST_ERR_LOG=$PWD/gpurun_out/err_log.txt python -m pytest tests -q -m gpu 2>&1 | tail -3
python tools/tol_report.py gpurun_out/err_log.txt gpurun_out/observed_errors.json | tail -40
