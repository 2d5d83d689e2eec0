"""Drop-in for the reference's ``jclip`` package on the MI355X HIP engine (see clip.py / model.py)."""
