"""Alias package: the reference import paths (model.*) resolve to t2ms_amd.model.*"""
