// srx_text.cpp -- the text front end that feeds the training path: pronunciation lexicon and pronouncer
// (the surface of src/pronounce.hpp:81-121 of the reference) and the PROMPTS-line utterance constructor
// (src/utterance.cpp:14-28).  Plain host string work: it decides WHICH tied states an utterance's HMM has, so
// it has to agree with the reference word by word, but nothing here is on the device path.
//
// Behaviour follows src/pronounce.cpp:94-278, including the parts that read like accidents, because they change
// the phone strings the acoustic model is trained on:
//   * an affix hit returns the STEM's first pronunciation only; the affix lexicons' own phones are never used
//     (src/pronounce.cpp:202-231);
//   * the two-way split puts a silence before BOTH halves: the pronlist overload of append_pron drops its
//     leading_silence argument (src/pronounce.cpp:166-168);
//   * a hyphenated word gets a doubled silence before its first and last piece (each piece already starts with
//     one, src/pronounce.cpp:190-198);
//   * the split search and the give-up `throw 1` sit inside the first turn of the prefix loop
//     (src/pronounce.cpp:213-255), so only the alphabetically first prefix is ever tried;
//   * a lexicon line that ends in blanks repeats its last phone (the extraction loop at src/pronounce.cpp:124-127
//     tests good() before reading; six lines of the reference's cmu.txt are like that).
// Where the reference runs off the end of a non-void function or indexes an empty string (a word of fewer than
// four letters that is not in the lexicon, an empty prefix lexicon, an empty word, an unknown phone name) there
// is no behaviour to match; this file throws 1 (the reference's "cannot pronounce" exception, which its caller
// catches per utterance, src/main.cpp:44-48), skips the empty word, or throws srx_error for the phone name.
#include <fstream>
#include <iostream>
#include <sstream>

#include "srx_host.hpp"

using std::string;
using std::vector;

typedef vector<phone::phone> phones_t;

pronlist::pronlist() {}
void pronlist::add(phones_t pr) { pronunciations.push_back(pr); }

pronlex::pronlex() {}

pronlex::pronlex(std::istream &in) {                                                     // src/pronounce.cpp:105-138
    if (in.fail()) {
        std::cout << "Failed to open pronunciation lexicon file " << std::endl;
        throw 1;  // the reference exits the process here; a library should not
    }
    string line;
    while (std::getline(in, line)) {
        if (!line.empty() && line[0] == ';') continue;
        std::istringstream is(line);
        string word, ph;
        is >> word;
        size_t paren = word.find('(');  // "WORD(2)" is another pronunciation of WORD
        if (paren != string::npos) word.erase(paren);
        phones_t p;
        while (is.good()) {
            string tok;
            if (is >> tok) ph = tok;    // a failed read (trailing blanks) leaves the previous name in place
            else if (ph.empty()) break;
            p.push_back(findPhone(ph));
        }
        add(word, p);
    }
}

void pronlex::add(string w, phones_t pron) { pr[w].add(pron); }

pronlist *pronlex::get(string w) {                                                       // src/pronounce.cpp:141-147
    std::map<string, pronlist>::iterator it = pr.find(w);
    if (it != pr.end()) return &it->second;
    if (srx::verbose()) std::cout << "!! didn't find " << w << std::endl;
    return nullptr;
}

pronouncer::pronouncer(const char *main_fn, const char *pre_fn, const char *suf_fn) {     // src/pronounce.cpp:150-156
    std::ifstream m_in(main_fn), p_in(pre_fn), s_in(suf_fn);
    main = pronlex(m_in);
    prefixes = pronlex(p_in);
    suffixes = pronlex(s_in);
    if (srx::verbose()) std::cout << "Got " << main.pr.size() << " words in main lexicon." << std::endl;
}

namespace {
void put(phones_t &res, const phones_t &wp, bool leading_silence) {
    if (leading_silence) res.push_back(phone::SIL);
    res.insert(res.end(), wp.begin(), wp.end());
}
bool ends_with(const string &w, const string &suf) { return w.size() >= suf.size() && w.compare(w.size() - suf.size(), suf.size(), suf) == 0; }
}  // namespace

phones_t pronouncer::pronounce_word(string w) {                                          // src/pronounce.cpp:178-257
    phones_t res;
    if (w.empty()) throw 1;
    if (w.back() == '-' || w.back() == '\'') w.pop_back();
    // every successful lookup ends the same way: silence, then the entry's first pronunciation
    auto found = [&](const string &key) -> bool {
        pronlist *p = main.get(key);
        if (!p) return false;
        put(res, p->pronunciations[0], true);
        return true;
    };
    if (found(w)) return res;
    if (w.size() < 4) throw 1;

    size_t dash = w.find('-');
    if (dash != string::npos) {
        bool first = true;
        for (; dash != string::npos; dash = w.find('-')) {
            put(res, pronounce_word(w.substr(0, dash)), first);
            w = w.substr(dash + 1);
            first = false;
        }
        put(res, pronounce_word(w), true);
        return res;
    }

    for (const auto &s : suffixes.pr)
        if (ends_with(w, s.first) && found(w.substr(0, w.size() - s.first.size()))) return res;

    if (!prefixes.pr.empty()) {
        const string &pre = prefixes.pr.begin()->first;
        if (pre.size() < w.size() && w.compare(0, pre.size(), pre) == 0) {
            const string rest = w.substr(pre.size());
            if (found(rest)) return res;
            for (const auto &s : suffixes.pr)
                if (ends_with(rest, s.first) && found(rest.substr(0, rest.size() - s.first.size()))) return res;
        }
    }

    // two dictionary words back to back: cut points walk outwards from the middle, left side first
    const size_t half = w.size() / 2;
    for (size_t split = 0; split + 1 < half; split++) {
        for (int side = -1; side <= 1; side += 2) {
            const size_t cut = side < 0 ? half - split : half + split;
            pronlist *p = main.get(w.substr(0, cut));
            pronlist *q = p ? main.get(w.substr(cut)) : nullptr;
            if (p && q) {
                put(res, p->pronunciations[0], true);
                put(res, q->pronunciations[0], true);
                return res;
            }
            if (split == 0) break;
        }
    }
    if (srx::verbose()) std::cout << "Warning! Cannot pronounce " << w << std::endl;
    throw 1;
}

phones_t pronouncer::pronounce(string s) {                                               // src/pronounce.cpp:260-278
    phones_t res;
    std::istringstream is(s);
    string w;
    while (is >> w) {
        phones_t wp = pronounce_word(w);
        res.insert(res.end(), wp.begin(), wp.end());
    }
    res.push_back(phone::SIL);
    return res;
}

utterance::utterance(std::istream &in, string path_base, acoustic_model &acm, pronouncer &pr) : hmm(nullptr) {
    string mfcc_fname;                                                                   // src/utterance.cpp:14-28
    in >> mfcc_fname;
    std::getline(in, text);
    pronunciation = pr.pronounce(text);
    hmm = new class hmm(pronunciation, &acm);
    features = read_features(path_base + mfcc_fname, 2);
    if (srx::verbose()) {
        std::cout << mfcc_fname << "\n\ttext: " << text << "\n\tpronunciation: ";
        for (phone::phone p : pronunciation) std::cout << phone::names[p] << " ";
        std::cout << std::endl;
    }
}
