#!/bin/bash
timeout 120 ./tools/micro/overlap
