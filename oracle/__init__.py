"""TEST INFRASTRUCTURE ONLY: the CPU restatements of the reference (checkers).  Nothing under jsmpeg_amd/ imports this package."""
