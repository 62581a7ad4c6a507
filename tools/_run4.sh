#!/bin/bash
for i in 1 2; do timeout 900 python tools/ubench/big_generic.py 2>&1 | grep "ms/step\|full"; done
