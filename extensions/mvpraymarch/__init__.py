"""Import-path compatibility with the reference tree (`extensions.*`): thin re-exports of ava-256_amd."""
