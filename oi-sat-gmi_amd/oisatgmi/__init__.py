"""placeholder (filled in below)"""
