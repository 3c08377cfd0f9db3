"""SVD completion passes on the HIP path (mirrors of model/SVD_2pass_prob_uncertain*.py)."""
