#!/bin/bash
timeout 300 python tools/vendor_reference.py 2>&1 | grep -v "^$" | tail -8
