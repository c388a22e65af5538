"""Alias package: `utils.*` re-exports avsum_amd.utils.* so that the reference's scripts, which import
`utils.*` and `src.utils.*`, run unmodified against the MI355X implementation."""
